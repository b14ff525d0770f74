"""Multi-rank runs of the product path ON THE GPU (HIP kernels, no cpu_port):
  * transport 'nccl' (= RCCL over xGMI, one GPU per rank): needs >= 2 visible GPUs, skipped otherwise;
  * transport 'gloo' with both ranks on cuda:0: runs on a 1-GPU box and exercises exactly the same host logic
    (FusedSyncBatchNorm's packed fp64 all-reduces, the cross-rank contrast set, DDP bucketing, bench.py's launcher);
    only the wire differs (RCCL refuses two ranks on one device).
Oracles: cross-rank loss == single-process loss on the concatenated global batch with
gradient/world == the single-process gradient slice; DDP + SyncBN replicas stay bit-identical and equal the
single-process run on the whole batch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need(backend):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: %d visible" % torch.cuda.device_count())


def _spawn(worker, world, *args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _init(rank, world, port, backend):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CSEG_DIST_BACKEND=backend)
    from contrastiveseg_amd.lib.utils.distributed import device_index, setup_process_group
    setup_process_group()
    torch.cuda.set_device(device_index())
    return torch.device("cuda", device_index())


def _loss_case():
    from oracle.make_golden import LOSS_CASES, case_inputs
    c = dict(LOSS_CASES["mid_self"])
    return c, case_inputs(c)


def _loss_cfg(c, cross_rank, budget="per_rank", rng="global", max_samples=256):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    k = dict(proj_dim=c["D"], base_temperature=0.07, use_rmi=False, warmup_iters=0)
    k.update(c["contrast"])
    k.update(max_samples=max_samples, cross_rank=cross_rank, cross_rank_budget=budget, cross_rank_rng=rng)
    return Configer(config_dict={"data": {"num_classes": c["K"]},
                                 "network": {"loss_weights": {"aux_loss": 0.4, "seg_loss": 1.0}}, "contrast": k,
                                 "loss": {"loss_type": "contrast_ce_loss",
                                          "params": {"ce_ignore_index": -1, "ce_reduction": "elementwise_mean"}}})


def _cross_rank_worker(rank, world, port, q, backend):
    import torch.distributed as dist
    dev = _init(rank, world, port, backend)
    from contrastiveseg_amd.lib.loss.loss_contrast import PixelContrastLoss
    c, (target, seg, embed, _) = _loss_case()
    B = c["B"] // world
    sl = slice(rank * B, (rank + 1) * B)
    crit = PixelContrastLoss(_loss_cfg(c, True))
    e = torch.from_numpy(embed[sl]).to(dev).requires_grad_(True)
    torch.manual_seed(11)
    loss = crit(e, torch.from_numpy(target[sl]).to(dev), seg=torch.from_numpy(seg[sl]).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    q.put((rank, float(loss.detach()), e.grad.cpu().numpy(), crit.last_selection["plan"].N))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_cross_rank_contrast_set_on_device(backend):
    _need(backend)
    world = 2
    res = _spawn(_cross_rank_worker, world, backend)
    from contrastiveseg_amd.lib.loss.loss_contrast import PixelContrastLoss
    dev = torch.device("cuda:0")
    c, (target, seg, embed, _) = _loss_case()
    crit = PixelContrastLoss(_loss_cfg(c, False, max_samples=256 * world))      # single process, whole batch
    e = torch.from_numpy(embed).to(dev).requires_grad_(True)
    torch.manual_seed(11)
    want = crit(e, torch.from_numpy(target).to(dev), seg=torch.from_numpy(seg).to(dev))
    want.backward()
    B = c["B"] // world
    for rank, loss, grad, n in res:
        assert n == crit.last_selection["plan"].N
        assert abs(loss - float(want)) <= 1e-5 * max(1.0, abs(float(want))), (loss, float(want))
        ref = e.grad.cpu().numpy()[rank * B:(rank + 1) * B]
        assert np.allclose(grad / world, ref, rtol=2e-4, atol=1e-8), np.abs(grad / world - ref).max()


def _trainer_cfg(global_batch):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    cfg = Configer(configs=os.path.join(ROOT, "configs", "cityscapes", "H_48_D_4.json"))
    cfg.update(["network", "backbone"], "hrnet18")
    cfg.update(["data", "num_classes"], 7)
    cfg.get("loss", "params").pop("ce_weight", None)
    cfg.update(["train", "batch_size"], global_batch)
    cfg.get("train", "data_transformer")["input_size"] = [256, 128]
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["contrast", "max_views"], 12)
    cfg.update(["contrast", "cross_rank_rng"], "global")
    cfg.update(["contrast", "cross_rank_budget"], "global")
    cfg.update(["solver", "max_iters"], 2)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    return cfg


def _batches(global_batch, n):
    g = torch.Generator().manual_seed(77)
    out = []
    for _ in range(n):
        img = torch.randn(global_batch, 3, 128, 256, generator=g)
        lab = torch.full((global_batch, 128, 256), -1, dtype=torch.long)
        for b in range(global_batch):
            lab[b] = int(torch.randint(0, 7, (1,), generator=g))
            for _r in range(10):
                cls = int(torch.randint(-1, 7, (1,), generator=g))
                y0, x0 = int(torch.randint(0, 128, (1,), generator=g)), int(torch.randint(0, 256, (1,), generator=g))
                lab[b, y0:y0 + 48, x0:x0 + 96] = cls
        out.append((img, lab))
    return out


PICK = ("backbone.conv1.weight", "backbone.bn1.running_var", "cls_head.3.weight", "proj_head.proj.2.weight")


def _ddp_worker(rank, world, port, q, backend):
    import torch.distributed as dist
    dev = _init(rank, world, port, backend)
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    torch.manual_seed(304)
    tr = Trainer(_trainer_cfg(4), train_loader=[])
    assert isinstance(tr.seg_net, torch.nn.parallel.DistributedDataParallel)
    for m in tr.seg_net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    tr.seg_net.train()
    B = 4 // world
    losses, picks = [], []
    for img, lab in _batches(4, 2):
        sl = slice(rank * B, (rank + 1) * B)
        losses.append(float(tr.train_step({"img": img[sl].to(dev), "labelmap": lab[sl].to(dev)})))
        sd = tr.seg_net.module.state_dict()
        picks.append({k: sd[k].detach().cpu().numpy().copy() for k in PICK})
    torch.cuda.synchronize()
    q.put((rank, losses, picks))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_ddp_syncbn_two_ranks_equal_single_process(backend):
    """2 ranks x 2 images, DDP + FusedSyncBatchNorm + cross-rank contrast set (budget/rng 'global') == one process with
    the 4 images: replicas bit-identical; weights after two SGD steps equal to the single-process run within the fp32
    noise of the different reduction orders; the CE term is the mean of per-rank means (reference DDP semantics), which
    on these labels differs from the global mean by < 1e-3 relative."""
    _need(backend)
    res = _spawn(_ddp_worker, 2, backend)
    for step in (0, 1):
        for k in PICK:
            assert np.array_equal(res[0][2][step][k], res[1][2][step][k]), "replicas diverged: " + k
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(304)
    tr = Trainer(_trainer_cfg(4), train_loader=[])
    for m in tr.seg_net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    tr.seg_net.train()
    w0 = {k: tr.seg_net.state_dict()[k].detach().cpu().numpy().copy() for k in PICK}
    single, after1 = [], None
    for img, lab in _batches(4, 2):
        single.append(float(tr.train_step({"img": img.to(dev), "labelmap": lab.to(dev)})))
        if after1 is None:
            after1 = {k: tr.seg_net.state_dict()[k].detach().cpu().numpy().copy() for k in PICK}
    # the displayed loss is rank-local in DDP: contrast term global (identical), CE term per-rank -> compare the mean
    ddp_mean = 0.5 * (res[0][1][0] + res[1][1][0])
    assert abs(ddp_mean - single[0]) <= 2e-3 * abs(single[0]), (ddp_mean, single)
    ddp1 = res[0][2][0]
    assert np.abs(ddp1["backbone.bn1.running_var"] - after1["backbone.bn1.running_var"]).max() <= 1e-5
    # first SGD update: DDP's averaged gradient == the single-process gradient up to the fp32 backward noise of this
    # network (1e-2 class, tests/test_step_golden.py) and the per-rank-mean CE normalisation
    for k in ("backbone.conv1.weight", "cls_head.3.weight", "proj_head.proj.2.weight"):
        d_ddp, d_one = ddp1[k] - w0[k], after1[k] - w0[k]
        err = np.linalg.norm(d_ddp - d_one) / np.linalg.norm(d_one)
        assert err <= 0.1, (k, err)
    assert all(np.isfinite(res[0][1])) and np.isfinite(single[1])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must start two ranks itself and print ONE JSON
    line from rank 0 (a driver-side `python bench.py --gpus 8` died on an assert). RCCL when two GPUs are
    visible, otherwise the gloo dry run with both ranks on cuda:0."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    backend = [] if torch.cuda.device_count() >= 2 else ["--backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-kernels", "--no-cpu-baseline", "--no-fp32-pass", "--global-batch", "2"] + backend,
                         env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 2
    if torch.cuda.device_count() >= 2:
        assert d["weak"] is not None and d["weak"]["global_batch"] == 4      # extra weak pass: RCCL runs only
    assert d["value"] > 0 and np.isfinite(d["config"]["final_loss"])


def _lockstep_worker(rank, world, port, q, backend, grouped):
    """One rank inside a process group with the multi-rank code paths forced on (CSEG_DIST_SINGLE_RANK=1): DDP wrapper, SyncBN with the
    branches in lockstep around the batched exchange, counts / anchor all-gathers -- the depth nodes on the grouped launches of round 6
    or on the per-member calls."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      CSEG_DIST_BACKEND=backend, CSEG_DIST_SINGLE_RANK="1", CSEG_BRANCH_STREAMS="0",
                      CSEG_BLOCK_GROUP="1" if grouped else "0",
                      # the per-member form on the grouped kernel's tile body (16-channel chunks) for every branch width
                      CSEG_CONV3X3_SB16_CH="48,96,192,384",
                      CSEG_BRANCH_STREAMS_MIN_PIXELS="1", CSEG_SB_MIN_TILES="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(304)
    cfg = _trainer_cfg(4)
    cfg.update(["network", "backbone"], "hrnet48")          # (HRNet-W18's 18 / 36 / 72 / 144 channels are outside the split kernels)
    tr = Trainer(cfg, train_loader=[])
    assert isinstance(tr.seg_net, torch.nn.parallel.DistributedDataParallel)
    assert HB.DDP_FORKS_OK, "the DDP wrapper must join the fork streams before its collectives"
    for m in tr.seg_net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    tr.seg_net.train()
    used = {"group": 0}
    run0 = K.conv3x3_group_run

    def run(items, want_stats=False):
        used["group"] += 1
        return run0(items, want_stats=want_stats)
    K.conv3x3_group_run = run
    losses, picks = [], []
    for img, lab in _batches(4, 2):
        losses.append(float(tr.train_step({"img": img.to(dev), "labelmap": lab.to(dev)})))
        sd = tr.seg_net.module.state_dict()
        picks.append({k: sd[k].detach().cpu().numpy().copy() for k in PICK})
    torch.cuda.synchronize()
    q.put((rank, losses, picks, used["group"]))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_depth_nodes_on_the_grouped_launches_equal_the_per_member_form():
    """Round 6: under a process group the residual blocks of a depth run as ONE node (fused_bn.BasicBlockGroupSync) on the GROUPED
    launches -- one kernel per pass for all branches, the statistics of a depth in one packed all-reduce. Against the same node on the
    per-member calls (CSEG_BLOCK_GROUP=0; both on the 16-channel-chunk tile body): the same arithmetic per output element, so the first
    loss agrees to rounding of the statistics records (each kernel sums a 64-pixel segment in its own fixed order) and the weights /
    BN buffers after the first SGD step to 1e-5 of their scale. (Round 5's opt-in that forked the lockstep convolutions onto side
    streams is gone: slower, and superseded by these launches.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _spawn(_lockstep_worker, 1, "gloo", False)[0]
    b = _spawn(_lockstep_worker, 1, "gloo", True)[0]
    assert a[3] == 0 and b[3] > 0, ("grouped launches taken", a[3], b[3])
    assert abs(a[1][0] - b[1][0]) <= 2e-5 * abs(a[1][0]), (a[1], b[1])
    for k in PICK:
        u, v = a[2][0][k], b[2][0][k]
        assert np.abs(u - v).max() <= 1e-5 * max(float(np.abs(u).max()), 1e-3), ("state after the first step differs: " + k, float(np.abs(u - v).max()))
