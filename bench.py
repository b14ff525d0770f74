"""bench.py -- contrastive train-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: when not already started by torch.distributed.run (no RANK/WORLD_SIZE in the environment) bench.py launches N
ranks of itself (one per GPU, RCCL over xGMI, rendezvous on 127.0.0.1) and forwards rank 0's JSON line; when the
driver starts it under `python -m torch.distributed.run ... bench.py --gpus N` it just joins the process group.

A "step" = one full contrastive training iteration of the workload's config: forward + criterion (HIP kernels) +
backward + RCCL gradient all-reduce + SGD, on a batch that is already resident in HBM. W untimed warm-up steps, then
exactly K timed steps bracketed by barrier + synchronize; time = MAX over ranks; rank 0 prints ONE JSON line.

Workloads (--workload):
  cfg2 (default)  BASELINE.json configs[1]/[2]: HRNet-W48 + contrast_ce_loss, synthetic Cityscapes 3x512x1024x19,
                  global batch 8 (`configs/cityscapes/H_48_D_4.json`), fp32, tau 0.1, max_samples 1024, with_embed on.
  cfg4            configs[3]: DeepLabV3-R101-d8 + contrast_auxce_loss, 3x512x1024x19, global batch 8.
  cfg4mem         the same network with the 4096-entry-per-class-pair pixel/segment memory bank (deeplab_v3_mem).
  cfg5            configs[4]: HRNet-W48-OCR + contrast_auxce_loss, COCO-Stuff shapes 3x520x520x171, global batch 16,
                  blocky labels (uniform labels cannot be mined at 171 classes).
Scaling (--scaling): `strong` (default) = the BASELINE configuration: GLOBAL batch fixed (8 resp. 16), split over the
ranks; `weak` = per-GPU batch fixed at the global batch of the config. With N > 1 and strong scaling the weak-scaling
throughput is measured as well (fewer steps, after the timed region) and reported under "weak".

The contract line is the LAST line of stdout and at most 2 KB (the driver keeps a bounded tail of stdout; BENCH_r03 lost its
head to a 12 KB line). It carries
  roofline      whole step against the roof of THIS implementation: achieved = images/s x TFLOP/image (BASELINE.md section 2,
                counted on the reference; HIP-event step time) / peak = the same flops with every pipe at its dense peak (split-
                operand convolutions at 2500 / 3 TFLOP/s for f16x3, the rest at the 157.3 TFLOP/s fp32 MFMA peak); frac = A / P;
                traffic = HBM bytes per step from the committed rocprofv3 PMC passes (profiles/*_step_pmc.json) or null;
                dominant_kernel = the split-operand kernel with the largest calls x time product of this run {name, frac, us}.
  cpu_baseline  N=1, rank 0 only: CPU port of the same train step (same model classes, oracle/cpu_port.py device
                half) timed on this box's usable host cores on a bounded sample, in a subprocess with a timeout.
Everything else -- per-kernel rooflines of every hand-written kernel (`kernels`), of every split-operand launch shape of the
step (`split_kernels`), the strict-fp32 comparison pass, notes -- goes to bench_detail.json (next to bench.py and under
gpurun_out/ when that exists).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fwd + loss + bwd TFLOP per image, counted on the reference
WORKLOADS = {
    "cfg2": dict(config="cityscapes/H_48_D_4.json", tflop=2.0759, batch=8, labels="uniform",
                 name="BASELINE.json configs[1]: HRNet-W48 + contrast_ce_loss, synthetic Cityscapes 3x512x1024x19, "
                      "tau=0.1, max_samples=1024, with_embed=True, SGD(0.01,0.9,5e-4)"),
    "cfg4": dict(config="cityscapes/R_101_D_8.json", tflop=4.8077, batch=8, labels="uniform",
                 name="BASELINE.json configs[3] (bank-free form): DeepLabV3-R101-d8 + contrast_auxce_loss, synthetic "
                      "Cityscapes 3x512x1024x19"),
    "cfg4mem": dict(config="cityscapes/R_101_D_8_MEM.json", tflop=4.8077, batch=8, labels="uniform",
                    name="BASELINE.json configs[3]: DeepLabV3-R101-d8 + mem_contrast_auxce_loss with the per-class "
                         "pixel/segment memory bank, synthetic Cityscapes 3x512x1024x19"),
    "cfg5": dict(config="coco_stuff/H_48_D_4.json", tflop=1.5418, batch=16, labels="blocky",
                 name="BASELINE.json configs[4]: HRNet-W48-OCR + contrast_auxce_loss, synthetic COCO-Stuff "
                      "3x520x520x171, blocky labels"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 / fp16 MFMA; a split-operand product costs six ("bf16x6") or three ("f16x3") MFMAs
MFMAS_PER_PRODUCT = {"bf16x6": 6.0, "f16x3": 3.0}


def peak_split_tflops(arith):
    return PEAK_BF16_MFMA_TFLOPS / MFMAS_PER_PRODUCT[arith]


PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                   help="strong (default, the BASELINE configuration): global batch fixed, split over the ranks; "
                        "weak: per-GPU batch = the config's global batch")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    p.add_argument("--config", default=None, help="override the workload's config file")
    p.add_argument("--global-batch", type=int, default=None, help="override the workload's global batch")
    p.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; gloo lets several "
                                                   "ranks share one GPU for a dry run)")
    p.add_argument("--no-weak", action="store_true", help="skip the extra weak-scaling measurement at N > 1")
    p.add_argument("--dist-single-rank", action="store_true",
                   help="N = 1 only: run inside a process group of ONE rank with CSEG_DIST_SINGLE_RANK=1, so that every multi-rank code "
                        "path (DDP wrapper and its bucketed all-reduce, the batched SyncBN exchange with the branches in lockstep, the "
                        "counts / anchor all-gathers) runs -- through RCCL on the default backend. What one rank of an N-GPU job pays per "
                        "step at the given --global-batch (= its per-GPU batch), minus the wire time: the measurable part of the "
                        "multi-GPU design on a 1-GPU box. NOT the BASELINE metric: reported as workload '<wl>+dist1'.")
    p.add_argument("--miopen-find", type=int, default=0,
                   help="cudnn.benchmark = MIOpen exhaustive find (a 20+ min warm-up on a fresh box); default off: "
                        "immediate mode + the tuned records shipped in contrastiveseg_amd/miopen_db")
    p.add_argument("--channels-last", type=int, default=0)
    p.add_argument("--conv-arith", choices=["default", "fp32", "split_bf16"], default="default",
                   help="3x3 convolutions of the head / 48-96 channel branches: fp32 (MIOpen + fp32-MFMA kernel) or "
                        "split-bf16 x6 on the BF16 matrix cores (fp32-class accuracy); default = the package default")
    p.add_argument("--no-fp32-pass", action="store_true",
                   help="skip the extra measurement of the same step with --conv-arith fp32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernels", action="store_true")
    p.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--in-child", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--labels", choices=["uniform", "blocky"], default=None)
    return p.parse_args()


def build_trainer(args, world, device, global_batch):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    wl = WORKLOADS[args.workload]
    cfg = Configer(configs=args.config or os.path.join(ROOT, "configs", wl["config"]))
    assert global_batch % world == 0, "global batch %d not divisible by %d ranks" % (global_batch, world)
    cfg.update(["train", "batch_size"], global_batch)
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["solver", "max_iters"], 10 ** 9)
    cfg.update(["solver", "display_iter"], 10 ** 9)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    cfg.add(["network", "channels_last"], bool(args.channels_last))
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    loader = SyntheticLoader(cfg, device, length=1, seed=304, mode=args.labels or wl["labels"], fixed=True)
    tr.seg_net.train()
    tr.pixel_loss.train()
    return tr, cfg, next(iter(loader))


def time_kernel(fn, iters=20, warm=5):
    """HIP-event time per call in us over an eager loop. Ops of a few tens of microseconds are bounded by the host's
    launch rate here (Python + ctypes + allocator, 20-50 us per call); their device time is in the rocprofv3 kernel
    statistics under profiles/ (hipGraph replay of these loops was tried and crashed inside the autograd engine on
    ROCm 7.2, so the loop stays eager)."""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(3):                               # best of three rounds: allocator / clock hiccups of a round are not the op
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / iters        # us
        best = t if best is None else min(best, t)
    return best


def kernel_rooflines(device, B, K=19, H=512, W=1024, stride=4, D=256):
    """HIP-event timings of the hand-written kernels at the workload's shapes; algorithmic bytes per DESIGN.md."""
    from contrastiveseg_amd import kernels as Kn
    from contrastiveseg_amd.lib.loss.anchor_sampling import plan_selection
    h, w = H // stride, W // stride
    P = h * w
    g = torch.Generator(device="cpu").manual_seed(1)
    seg = torch.randn(B, K, h, w, generator=g).to(device)
    target = torch.randint(-1, K, (B, H, W), generator=g).to(device)
    embed = torch.nn.functional.normalize(torch.randn(B, D, h, w, generator=g), dim=1).to(device)
    out = {}

    def entry(name, us, bytes_=None, flops=None):
        e = {"us": round(us, 2)}
        if bytes_ is not None:
            gbs = bytes_ / us * 1e-3
            e.update(bound="hbm", bytes=int(bytes_), achieved_GBs=round(gbs, 1), frac=round(gbs / PEAK_HBM_GBS, 4))
        if flops is not None:
            tf = flops / us * 1e-6
            e.update(bound="mfma", flops=int(flops), achieved_TFLOPs=round(tf, 2),
                     frac=round(tf / PEAK_FP32_MFMA_TFLOPS, 4))
        out[name] = e

    # mining: seg + strided labels in, key/part_idx/counts out
    us = time_kernel(lambda: Kn.classify_partition(target, -1, seg=seg))
    entry("classify_partition", us, bytes_=B * K * P * 4 + B * P * 8 + B * P * (2 + 4))
    cp = Kn.classify_partition(target, -1, seg=seg)
    plan = plan_selection(cp["counts"].cpu().numpy(), 1024, 100)
    sel_pos = torch.from_numpy(plan.row_img * P + plan.row_off).to(device)
    a_lab = torch.from_numpy(plan.row_lab).to(device)
    N = plan.N
    emb = embed.clone().requires_grad_(True)

    def contrast_fwd():
        return Kn.PixelContrast.apply(emb, cp["part_idx"], sel_pos, a_lab, "self", 0.1, 0.07, None, None)[0]
    us_f = time_kernel(contrast_fwd)
    loss = contrast_fwd()
    us_fb = time_kernel(lambda: torch.autograd.grad(contrast_fwd(), emb))
    entry("contrast_self_fwd(gather+S+rows+mean) N=%d" % N, us_f, flops=2.0 * N * N * D)
    entry("contrast_self_fwd+bwd(+scatter,+268MB memset) N=%d" % N, us_fb, flops=2.0 * N * N * D * 2)
    # headline bank shape: 1024 anchors x 4096-entry bank (4104 with the class-0 quirk)
    A = torch.nn.functional.normalize(torch.randn(1024, D, generator=g), dim=1).to(device).requires_grad_(True)
    yl = torch.randint(0, K, (1024,), generator=g).int().to(device)
    sq = torch.nn.functional.normalize(torch.randn(K, 108, D, generator=g), dim=2).to(device)
    pq = torch.nn.functional.normalize(torch.randn(K, 108, D, generator=g), dim=2).to(device)

    def bank():
        return Kn.ContrastOnAnchors.apply(A, yl, "bank", 0.1, 0.07, None, None, sq, pq)
    entry("contrast_bank_fwd 1024x4104", time_kernel(bank), flops=2.0 * 1024 * 4104 * D)
    entry("contrast_bank_fwd+bwd 1024x4104", time_kernel(lambda: torch.autograd.grad(bank(), A)),
          flops=2.0 * 1024 * 4104 * D * 2)
    # the forward alone, both implementations: three launches with S through HBM vs the fused single launch.
    # Host + device time of back-to-back calls as the step issues them; device-only durations: profiles/r05_contrast_fused_kernels.txt
    anchors_self = Kn.gather_anchors(embed, cp["part_idx"], sel_pos)[0]
    was = Kn.CONTRAST_FUSED
    try:
        for tag, flag in (("three launches, S via HBM", "0"), ("fused, one launch", "1")):
            Kn.CONTRAST_FUSED = flag
            d_self = Kn._desc(0, anchors_self, a_lab, 0.1, 0.07)
            entry("contrast_self_fwd only [%s] N=%d" % (tag, N), time_kernel(lambda: Kn.contrast_forward(d_self, device)), flops=2.0 * N * N * D)
            d_bank = Kn._desc(2, A.detach(), yl, 0.1, 0.07, None, None, sq, pq)
            entry("contrast_bank_fwd only [%s] 1024x4104" % tag, time_kernel(lambda: Kn.contrast_forward(d_bank, device)),
                  flops=2.0 * 1024 * 4104 * D)
    finally:
        Kn.CONTRAST_FUSED = was
    # HRNet head input
    chans = [48, 96, 192, 384]
    feats = [torch.randn(B, c, h >> i, w >> i, generator=g).to(device).requires_grad_(True)
             for i, c in enumerate(chans)]
    in_b = sum(f.numel() for f in feats) * 4
    out_b = B * sum(chans) * P * 4
    entry("upcat_fwd", time_kernel(lambda: Kn.upsample_concat(feats)), bytes_=in_b + out_b)
    o = Kn.upsample_concat(feats)
    go = torch.randn_like(o)
    entry("upcat_bwd", time_kernel(lambda: torch.autograd.grad(o, feats, go, retain_graph=True)), bytes_=in_b + out_b)
    del o, go, feats
    # HRNet exchange unit, finest branch of a 4-branch module: 1 same-resolution term + 3 coarse terms, 48 channels
    same = [torch.randn(B, 48, h, w, generator=g).to(device).requires_grad_(True)]
    low = [torch.randn(B, 48, h >> i, w >> i, generator=g).to(device).requires_grad_(True) for i in (1, 2, 3)]
    fb = (same[0].numel() * 2 + sum(t.numel() for t in low)) * 4
    entry("fuse_sum_relu_fwd 48ch", time_kernel(lambda: Kn.fuse_sum_relu(same, low)), bytes_=fb)
    fo = Kn.fuse_sum_relu(same, low)
    fg = torch.randn_like(fo)
    entry("fuse_sum_relu_bwd 48ch", time_kernel(lambda: torch.autograd.grad(fo, same + low, fg, retain_graph=True)),
          bytes_=same[0].numel() * 4 * 3 + sum(t.numel() for t in low) * 4)
    del fo, fg, same, low
    # segmentation term
    wt = torch.ones(K, device=device)
    sg = seg.clone().requires_grad_(True)
    entry("upsample_ce_fwd", time_kernel(lambda: Kn.upsample_ce(sg, target, wt, -1)),
          bytes_=B * K * P * 4 + B * H * W * (8 + 4))              # seg + labels in, log-sum-exp out
    l = Kn.upsample_ce(sg, target, wt, -1)
    entry("upsample_ce_bwd", time_kernel(lambda: torch.autograd.grad(l, sg, retain_graph=True)),
          bytes_=2 * B * K * P * 4 + B * H * W * (8 + 4))
    del sg, l
    # fused BatchNorm (+ReLU / +residual): the two dominant activation shapes of the HRNet-W48 step
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d
    for C, hh, ww in ((48, h, w), (720, h, w)):
        bn = FusedBatchNorm2d(C).to(device).train()
        x = torch.randn(B, C, hh, ww, generator=g).to(device).requires_grad_(True)
        r = torch.randn(B, C, hh, ww, generator=g).to(device).requires_grad_(True)
        nb = x.numel() * 4
        entry("bn_relu_fwd %dch (stats + apply)" % C, time_kernel(lambda: bn(x, relu=True)), bytes_=3 * nb)
        y = bn(x, relu=True)
        gy = torch.randn_like(y)
        entry("bn_relu_bwd %dch (reduce + apply)" % C,
              time_kernel(lambda: torch.autograd.grad(y, x, gy, retain_graph=True)), bytes_=5 * nb)
        if C == 48:
            entry("bn_add_relu_fwd %dch" % C, time_kernel(lambda: bn(x, residual=r, relu=True)), bytes_=4 * nb)
            y2 = bn(x, residual=r, relu=True)
            entry("bn_add_relu_bwd %dch" % C,
                  time_kernel(lambda: torch.autograd.grad(y2, (x, r), gy, retain_graph=True)), bytes_=7 * nb)
            del y2
        del bn, x, r, y, gy
    # the classifier (720 -> K, csrc/cls1x1.hip): three streams over the 720-channel activation; algorithmic bytes = the wide tensor once
    # (+ the K-channel tensor), DESIGN.md section 13.13
    xc = torch.randn(B, 720, h, w, generator=g).relu_().to(device).requires_grad_(True)
    wc = (torch.randn(K, 720, 1, 1, generator=g) / 720 ** 0.5).to(device).requires_grad_(True)
    mk = (torch.rand(B, 720, 1, 1, generator=g) > 0.1).float().div_(0.9).to(device)
    wide, narrow = xc.numel() * 4, B * K * P * 4
    if Kn.cls1x1_eligible(xc, wc):
        import ctypes
        from contrastiveseg_amd import _hip
        wtc = Kn.cls1x1_weights(wc, B, mk).detach().contiguous()
        KPc = wtc.shape[2]
        xd = xc.detach()
        yc = torch.empty(B, K, h, w, device=device)
        gc = torch.randn(B, K, h, w, generator=g).to(device)
        dxc = torch.empty_like(xd)
        wsn = _hip.lib().cseg_cls1x1_wrw_ws_floats(B, 720, KPc, ctypes.c_long(P))
        wsc, dwc = torch.empty(wsn, device=device), torch.empty(B, 720, KPc, device=device)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        # (the C-ABI entry points with caller-owned buffers: through autograd.grad the host, not the kernel, sets the pace of a timing loop)
        entry("cls1x1_fwd 720->%d" % K, time_kernel(lambda: _hip.call("cseg_cls1x1_fwd", ptr(xd), ptr(wtc), None, B, 720, K, KPc,
                                                                    ctypes.c_long(P), ptr(yc), _hip.stream_ptr())), bytes_=wide + narrow)
        entry("cls1x1_bwd 720->%d (backward-data)" % K,
              time_kernel(lambda: _hip.call("cseg_cls1x1_bwd", ptr(gc), ptr(wtc), B, 720, K, KPc, ctypes.c_long(P), ptr(dxc),
                                            _hip.stream_ptr())), bytes_=wide + narrow)
        entry("cls1x1_wrw 720->%d (weight gradient)" % K,
              time_kernel(lambda: _hip.call("cseg_cls1x1_wrw", ptr(xd), ptr(gc), B, 720, K, KPc, ctypes.c_long(P), ptr(wsc), ptr(dwc),
                                            _hip.stream_ptr())), bytes_=wide + narrow)
        del wtc, yc, gc, dxc, wsc, dwc, xd
    del xc, wc, mk
    # the head's 3x3 convolution in the current split arithmetic (forward = backward-data): fp32-equivalent flops vs 2500 / k TF/s
    C = 720
    xh = torch.randn(B, C, h, w, device=device)
    wh = torch.randn(C, C, 3, 3, device=device) / (3.0 * C ** 0.5)
    axh = Kn.tensor_amax(xh) if Kn.split_arith_id() else None
    us = time_kernel(lambda: Kn.conv3x3_sb_run(xh, wh, False, ax=axh), iters=5, warm=2)
    fl = 2.0 * B * h * w * C * C * 9
    pk = peak_split_tflops(Kn.SPLIT_ARITH)
    out["conv3x3_split 720->720 (%s)" % Kn.SPLIT_ARITH] = {
        "us": round(us, 1), "bound": "mfma", "flops": int(fl), "achieved_TFLOPs": round(fl / us * 1e-6, 1),
        "peak_TFLOPs": round(pk, 1), "frac": round(fl / us * 1e-6 / pk, 4),
        "note": "fp32-equivalent flops; %d MFMAs per product; weights packed once per optimizer step (cached); MIOpen fp32 at "
                "this shape: 19.8 ms" % MFMAS_PER_PRODUCT[Kn.SPLIT_ARITH]}
    del xh, wh
    return out


def conv_arith_note(Kn, split_on):
    """config.conv3x3_arithmetic of the JSON line: which convolutions run in which arithmetic in this process."""
    if not split_on:
        return "fp32"
    how = {"f16x3": "split-fp16 x3 on the matrix cores (fp32 in/out; every operand = two fp16 pieces of the tensor scaled by a "
                    "power of two taken from max|tensor|, three fp16 piece products per fp32 product, fp32 accumulate",
           "bf16x6": "split-bf16 x6 on the matrix cores (fp32 in/out; every operand = three bf16 pieces, six bf16 piece products "
                     "per fp32 product, fp32 accumulate"}[Kn.SPLIT_ARITH]
    wrw = ""
    if Kn.CONV3X3_SB_WRW:
        wrw = " + weight gradient (%s channels)" % "/".join(str(c) for c in Kn.CONV3X3_SB_WRW_CHANNELS)
    one = ""
    if Kn.CONV1X1_SPLIT_BF16:
        one = "; 1x1 convolutions with tiling channel counts (projection head, bottlenecks) forward + backward-data" + (
            " + weight gradient (>= %d channels)" % Kn.CONV1X1_SB_WRW_MIN_CH if Kn.CONV1X1_SB_WRW else "")
    s2 = ""
    if getattr(Kn, "CONV3X3_S2_SPLIT", False) and Kn.SPLIT_ARITH == "f16x3":
        s2 = "; the 3x3 stride-2 convolutions of the fuse / transition layers in all three directions"
    return ("%s; fp32-class accuracy: whole-network logits vs fp64 5.1e-5 (f16x3) / 2.1e-5 (bf16x6) against fp32's own 4.3e-5, "
            "tools/split_bf16_probe.py, tests/test_gpu_conv3x3_sb.py) for the 720->720 head convolution and the %s-channel "
            "branches, forward + backward-data%s%s%s; everything else fp32 (MIOpen / rocBLAS)"
            % (how, "/".join(str(c) for c in Kn.CONV3X3_SB_BRANCH_CHANNELS), wrw, one, s2))


SPLIT_OPS = ("conv3x3_sb_run", "conv1x1_sb_run", "conv3x3_sb_wrw", "conv1x1_sb_wrw",
             "conv3x3_s2_run", "conv3x3_s2_bwd_run", "conv3x3_s2_wrw")     # the last three: stride 2 (round 3)


GROUP_OPS = ("conv3x3_group_run", "conv3x3_group_wrw")     # round 6: one launch over the parallel branches of a depth (kernels.BasicBlockGroup)


def _group_flops(name, members):
    """fp32-equivalent flops of one grouped launch: the sum over its member convolutions (3x3, stride 1)."""
    if name == "conv3x3_group_run":
        return sum(2.0 * sx[0] * sx[2] * sx[3] * sw[0] * sw[1] * 9 for sx, sw, _flip, _add in members)
    return sum(2.0 * sx[0] * sx[2] * sx[3] * sx[1] * sdy[1] * 9 for sx, sdy in members)


def tally_split_calls(tr, batch, Kn):
    """One extra (untimed) train step with counting wrappers around the four split-operand entry points of
    contrastiveseg_amd.kernels: {(op, operand shapes, flag): calls per step}. What the step REALLY launches -- the dominant
    kernel is picked from this and from live timings, not from a table written by hand."""
    counts = {}
    ops = SPLIT_OPS + tuple(n for n in GROUP_OPS if hasattr(Kn, n))
    saved = {n: getattr(Kn, n) for n in ops}

    def wrap(name, fn):
        def inner(*a, **k):
            if name == "conv3x3_group_run":         # items: (x, weight, transpose_flip, max|x| record, addend)
                members = tuple((tuple(x.shape), tuple(w.shape), bool(flip), add is not None) for x, w, flip, _ax, add in a[0])
                key = (name, members, (), bool(k.get("want_stats", a[1] if len(a) > 1 else False)))
            elif name == "conv3x3_group_wrw":       # items: (x, dy, max|x|, max|dy|)
                key = (name, tuple((tuple(x.shape), tuple(dy.shape)) for x, dy, _ax, _ady in a[0]), (), False)
            else:
                flag = bool(a[2]) if (name.endswith("_run") and len(a) > 2) else False
                key = (name, tuple(a[0].shape), tuple(a[1].shape), flag)
            counts[key] = counts.get(key, 0) + 1
            return fn(*a, **k)
        return inner
    try:
        for n in ops:
            setattr(Kn, n, wrap(n, saved[n]))
        tr.train_step(batch)
        torch.cuda.synchronize()
    finally:
        for n in ops:
            setattr(Kn, n, saved[n])
    return counts


def split_flops_of(counts):
    """fp32-equivalent flops per step on the split-operand kernels, from the tally alone (no timing)."""
    total = 0.0
    for (name, sa, sb, flag), n in counts.items():
        if name in GROUP_OPS:
            flops = _group_flops(name, sa)
        elif name.startswith("conv3x3_s2_"):
            if name == "conv3x3_s2_wrw":
                ci, co, ho, wo = sa[1], sb[1], sb[2], sb[3]
            else:
                ci, co = sb[1], sb[0]
                ho, wo = (sa[2], sa[3]) if name == "conv3x3_s2_bwd_run" else (sa[2] // 2, sa[3] // 2)
            flops = 2.0 * sa[0] * ho * wo * ci * co * 9
        elif name.endswith("_run"):
            flops = 2.0 * sa[0] * sa[2] * sa[3] * sb[0] * sb[1] * sb[2] * sb[3]
        else:
            flops = 2.0 * sa[0] * sa[2] * sa[3] * sa[1] * sb[1] * (9 if name == "conv3x3_sb_wrw" else 1)
        total += n * flops
    return total


def split_kernel_rooflines(counts, device, Kn):
    """Times every distinct (op, shape) of the tally in isolation (HIP events over a loop, weights packed and max|.| words
    computed once, as inside a step) -> list of dicts sorted by their share of the step, plus the fp32-equivalent flops per
    step that run on the split-operand kernels."""
    peak = peak_split_tflops(Kn.SPLIT_ARITH)
    g = torch.Generator(device="cpu").manual_seed(7)
    rows, split_flops = [], 0.0
    for (name, sa, sb, flag), n in sorted(counts.items(), key=lambda kv: -kv[1]):
        if name in GROUP_OPS:                       # a grouped launch: its members' operands, one call
            flops = _group_flops(name, sa)
            if name == "conv3x3_group_run":
                items = []
                for sx, sw, flip, add in sa:
                    x = torch.randn(*sx, generator=g).to(device)
                    w = (torch.randn(*sw, generator=g) / (9 * sw[1]) ** 0.5).to(device)
                    y_shape = (sx[0], sw[1] if flip else sw[0], sx[2], sx[3])
                    items.append((x, w, flip, Kn.tensor_amax(x), torch.randn(*y_shape, generator=g).to(device) if add else None))
                fn = lambda: Kn.conv3x3_group_run(items, want_stats=flag)
                kind = "backward-data" if sa[0][2] else ("forward + statistics" if flag else "forward")
                what = "grouped conv3x3 %s: %s" % (kind, " + ".join("%d@%dx%dx%d" % (sx[1], sx[0], sx[2], sx[3]) for sx, _, _, _ in sa))
            else:
                items = []
                for sx, sdy in sa:
                    x = torch.randn(*sx, generator=g).to(device)
                    dy = (torch.randn(*sdy, generator=g) * 1e-3).to(device)
                    items.append((x, dy, Kn.tensor_amax(x), Kn.tensor_amax(dy)))
                fn = lambda: Kn.conv3x3_group_wrw(items)
                what = "grouped conv3x3 weight gradients: %s" % " + ".join("%d@%dx%dx%d" % (sx[1], sx[0], sx[2], sx[3]) for sx, _ in sa)
            us = time_kernel(fn, iters=20, warm=2)
            tf = flops / us * 1e-6
            rows.append({"kernel": what, "entry": name, "calls_per_step": n, "us_per_launch": round(us, 1),
                         "ms_per_step": round(n * us * 1e-3, 3), "algorithmic_flops_per_launch": int(flops),
                         "achieved_TFLOPs": round(tf, 1), "peak_TFLOPs": round(peak, 1), "frac": round(tf / peak, 4)})
            split_flops += n * flops
            del items
            continue
        a = torch.randn(*sa, generator=g).to(device)
        if name.startswith("conv3x3_s2_"):
            if name == "conv3x3_s2_wrw":
                dy = (torch.randn(*sb, generator=g) * 1e-3).to(device)
                ax, ad = Kn.tensor_amax(a), Kn.tensor_amax(dy)
                fn = lambda: Kn.conv3x3_s2_wrw(a, dy, ax=ax, ady=ad)
                ci, co, ho, wo, kind = sa[1], sb[1], sb[2], sb[3], "weight gradient"
            else:
                w = (torch.randn(*sb, generator=g) / (9 * sb[1]) ** 0.5).to(device)
                ax = Kn.tensor_amax(a)
                bwd = name == "conv3x3_s2_bwd_run"
                fn = (lambda: Kn.conv3x3_s2_bwd_run(a, w, ady=ax)) if bwd else (lambda: Kn.conv3x3_s2_run(a, w, ax=ax))
                ci, co, kind = sb[1], sb[0], "backward-data" if bwd else "forward"
                ho, wo = (sa[2], sa[3]) if bwd else (sa[2] // 2, sa[3] // 2)
            flops = 2.0 * sa[0] * ho * wo * ci * co * 9
            what = "conv3x3 stride 2 %d->%d %s @%dx%dx%d (output)" % (ci, co, kind, sa[0], ho, wo)
        elif name.endswith("_run"):
            w = (torch.randn(*sb, generator=g) / (sb[1] * sb[2] * sb[3]) ** 0.5).to(device)
            taps = sb[2] * sb[3]
            nt = Kn.conv3x3_sb_pick_nt(a, sb[1] if flag else sb[0]) if (taps == 9 and sb[0] in Kn.CONV3X3_SB_PICK_NT_CHANNELS) else 0
            ax = Kn.tensor_amax(a) if Kn.split_arith_id() else None
            if taps == 9:
                fn = lambda: Kn.conv3x3_sb_run(a, w, flag, None, nt, ax=ax)
            else:
                fn = lambda: Kn.conv1x1_sb_run(a, w, flag, None, ax=ax)
            flops = 2.0 * sa[0] * sa[2] * sa[3] * sb[0] * sb[1] * taps
            what = "%s %d->%d %s @%dx%dx%d" % ("conv3x3" if taps == 9 else "conv1x1", sb[0] if flag else sb[1],
                                               sb[1] if flag else sb[0], "backward-data" if flag else "forward", sa[0], sa[2], sa[3])
        else:
            dy = (torch.randn(*sb, generator=g) * 1e-3).to(device)
            ax, ad = (Kn.tensor_amax(a), Kn.tensor_amax(dy)) if Kn.split_arith_id() else (None, None)
            taps = 9 if name == "conv3x3_sb_wrw" else 1
            fn = (lambda: Kn.conv3x3_sb_wrw(a, dy, ax=ax, ady=ad)) if taps == 9 else (lambda: Kn.conv1x1_sb_wrw(a, dy, ax=ax, ady=ad))
            flops = 2.0 * sa[0] * sa[2] * sa[3] * sa[1] * sb[1] * taps
            what = "%s %d->%d weight gradient @%dx%dx%d" % ("conv3x3" if taps == 9 else "conv1x1", sa[1], sb[1], sa[0], sa[2], sa[3])
        us = time_kernel(fn, iters=5 if flops > 5e11 else 20, warm=2)
        tf = flops / us * 1e-6
        rows.append({"kernel": what, "entry": name, "calls_per_step": n, "us_per_launch": round(us, 1),
                     "ms_per_step": round(n * us * 1e-3, 3), "algorithmic_flops_per_launch": int(flops),
                     "achieved_TFLOPs": round(tf, 1), "peak_TFLOPs": round(peak, 1), "frac": round(tf / peak, 4)})
        split_flops += n * flops
        del a
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows, split_flops


FAMILIES = {"conv3x3_group_run": "grouped 3x3 forward / backward-data over the parallel branches of a depth (conv3x3_group_pc12_kernel)",
            "conv3x3_group_wrw": "grouped 3x3 weight gradients (conv3x3_wrw2_group_kernel + sb_wrw_reduce_group_kernel)",
            "conv3x3_sb_wrw": "3x3 weight gradients (conv3x3_sb_wrw2_kernel + sb_wrw_reduce_kernel)",
            "conv3x3_sb_run": "3x3 stride-1 forward / backward-data (conv3x3_sb16r/sb16p/sb/sb16/sb8 kernels)",
            "conv1x1_sb_run": "1x1 forward / backward-data (conv1x1_sb_kernel)", "conv1x1_sb_wrw": "1x1 weight gradients (conv1x1_sb_wrw_kernel)",
            "conv3x3_s2_run": "3x3 stride-2 forward", "conv3x3_s2_bwd_run": "3x3 stride-2 backward-data", "conv3x3_s2_wrw": "3x3 stride-2 weight gradients"}


def kernel_families(split_rows):
    """The per-(op, shape) rows summed per kernel FAMILY -- one entry point of contrastiveseg_amd.kernels = the device kernels
    rocprofv3 lists under one name pattern: flops and time over ALL launches of a step, so that `frac` is what the step pays for the
    family, not what its best launch reaches (the line used to show the 720-channel head launch alone, 0.46,
    while the 184 small launches of the same kernel ran at 0.27)."""
    fam = {}
    for r in split_rows or []:
        f = fam.setdefault(r["entry"], {"entry": r["entry"], "name": FAMILIES.get(r["entry"], r["entry"]), "calls_per_step": 0,
                                        "ms_per_step": 0.0, "flops_per_step": 0.0, "peak_TFLOPs": r["peak_TFLOPs"], "members": []})
        f["calls_per_step"] += r["calls_per_step"]
        f["ms_per_step"] += r["ms_per_step"]
        f["flops_per_step"] += r["calls_per_step"] * r["algorithmic_flops_per_launch"]
        f["members"].append({"kernel": r["kernel"], "calls_per_step": r["calls_per_step"], "us_per_launch": r["us_per_launch"],
                             "frac": r["frac"]})
    out = []
    for f in fam.values():
        f["achieved_TFLOPs"] = round(f["flops_per_step"] / max(f["ms_per_step"], 1e-9) * 1e-9, 1)
        f["frac"] = round(f["achieved_TFLOPs"] / f["peak_TFLOPs"], 4)
        f["ms_per_step"] = round(f["ms_per_step"], 3)
        out.append(f)
    out.sort(key=lambda f: -f["ms_per_step"])
    return out


def dominant_kernel(split_rows, arith):
    """roofline.dominant_kernel: the split-operand kernel FAMILY with the largest time per step of THIS run (isolated HIP-event
    timings of every (op, shape) the step launches x its calls per step), flops and time summed over the family's launches."""
    fams = kernel_families(split_rows)
    if not fams:
        return None
    d = fams[0]
    worst = min(d["members"], key=lambda m: m["frac"])
    return {"name": d["name"], "entry": d["entry"], "bound": "mfma", "achieved": d["achieved_TFLOPs"], "peak": d["peak_TFLOPs"],
            "unit": "TFLOP/s fp32-equivalent (%s: %d MFMAs per product, peak = 2500 / %d)"
                    % (arith, MFMAS_PER_PRODUCT[arith], MFMAS_PER_PRODUCT[arith]),
            "frac": d["frac"], "us_per_launch": round(d["ms_per_step"] * 1e3 / max(d["calls_per_step"], 1), 1),
            "calls_per_step": d["calls_per_step"], "ms_per_step": d["ms_per_step"],
            "algorithmic_flops_per_step": int(d["flops_per_step"]), "worst_member": worst, "members": d["members"],
            "picked_from": "live tally of one train step x live HIP-event timings in isolation (bench.py:split_kernel_rooflines), summed "
                           "per entry point; in-step durations: profiles/r06_step_steady_kernel_stats.csv"}


LINE_LIMIT = 2048          # bytes of the contract line: the driver keeps a bounded tail of stdout (BENCH_r03: parsed null at ~12 KB)
DETAIL_NAME = "bench_detail.json"


def arithmetic_name(Kn, split_on):
    """`dtype` of the contract line: the arithmetic type the dominant path computes in."""
    if not split_on:
        return "fp32"
    arith = getattr(Kn, "SPLIT_ARITH", "bf16x6")
    return {"f16x3": "fp32 in/out, f16x3 split MFMA (fp32 accumulate)",
            "bf16x6": "fp32 in/out, bf16x6 split MFMA (fp32 accumulate)"}[arith]


def assemble_line(args, wl, cfg, world, global_batch, dt, ev_ms, final_loss, split_on, Kn, backend, fp32_pass, weak, cpu,
                  kernels, split_rows=None, split_flops=0.0):
    """-> (line, detail). `line` = the ONE JSON line of the contract, at most LINE_LIMIT bytes, printed LAST; `detail` = everything
    else (per-kernel tables, notes, the comparison passes), written to bench_detail.json. Pure function of the measurements:
    exercised on CPU by tests/test_host_surface.py so that a formatting slip cannot cost a GPU run its result.

    roofline (top level) = the whole step against the roof of THIS implementation: achieved = images/s x TFLOP/image of
    fp32-equivalent work (HIP-event step time); peak = the rate at which the same flops would run with every pipe at its dense
    peak -- split-operand flops at 2500 / (MFMAs per product) TFLOP/s, the rest at the 157.3 TFLOP/s fp32 MFMA peak -- so
    frac = achieved / peak = roof time / measured time. The figure against the fp32 MFMA peak alone (what a pure fp32
    implementation is bounded by; can exceed 1) is kept as roofline.vs_fp32_mfma_peak."""
    ips = global_batch * args.steps / dt
    step_ms = ev_ms / args.steps
    ips_ev = global_batch * args.steps / (ev_ms * 1e-3)
    achieved = ips_ev * wl["tflop"]
    arith = getattr(Kn, "SPLIT_ARITH", "bf16x6")
    total = wl["tflop"] * 1e12 * global_batch                      # flops per step, all ranks
    split = min(split_flops * world, total) if split_on else 0.0   # the tally is per rank
    rest = total - split
    roof_s = (split / (peak_split_tflops(arith) * 1e12) + rest / (PEAK_FP32_MFMA_TFLOPS * 1e12)) / world
    peak = total / roof_s * 1e-12                                   # TFLOP/s, whole job
    traffic, traffic_src = step_traffic() if (world == 1 and args.workload == "cfg2") else (None, None)
    W, H = cfg.get("train", "data_transformer")["input_size"]
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "split_TFLOP_per_step": round(split * 1e-12, 3), "fp32_TFLOP_per_step": round(rest * 1e-12, 3),
                "vs_fp32_mfma_peak": round(achieved / (PEAK_FP32_MFMA_TFLOPS * world), 4)}
    dom = dominant_kernel(split_rows, arith) if (split_on and split_rows) else None
    if dom is not None:
        roofline["dominant_kernel"] = {"name": dom["name"][:90], "frac": dom["frac"], "us": dom["us_per_launch"],
                                       "achieved": dom["achieved"], "peak": dom["peak"], "calls_per_step": dom["calls_per_step"],
                                       "ms_per_step": dom["ms_per_step"]}
    if traffic_src:
        roofline["traffic_source"] = ("committed rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE runs): " if traffic is not None else "") + traffic_src[:110]
    cpu_short = None
    if cpu is not None:
        cpu_short = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "error") if k in cpu}
        if "sample" in cpu:
            cpu_short["sample"] = cpu["sample"][:160]
    line = {
        "metric": "images/sec contrastive train step, HRNet-W48 1024x512 bs8" if (args.workload == "cfg2" and not getattr(args, "dist_single_rank", False)) else
                  "images/sec contrastive train step, " + args.workload + ("+dist1" if getattr(args, "dist_single_rank", False) else ""),
        "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": arithmetic_name(Kn, split_on), "data": "synthetic",
        "config": {"workload": args.workload + ": " + wl["name"][:150], "model": cfg.get("network", "model_name"),
                   "loss": cfg.get("loss", "loss_type"), "global_batch": global_batch,
                   "per_gpu_batch": global_batch // world, "input": [3, H, W],
                   "arithmetic": ("%s on the matrix cores for the 3x3 / 1x1 convolutions, rest fp32" % arith) if split_on else "fp32",
                   "parallelism": "dp%d" % world if not getattr(args, "dist_single_rank", False) else
                                  "dp1 inside a one-rank process group: multi-rank code paths on (DDP, SyncBN exchange, all-gathers)",
                   "backend": backend,
                   "step_graph": os.environ.get("CSEG_STEP_GRAPH_STATE"),
                   "final_loss": round(final_loss, 5)},
        "roofline": roofline, "cpu_baseline": cpu_short, "detail": DETAIL_NAME,
    }
    if fp32_pass is not None and "ms_per_step" in fp32_pass:
        line["fp32_conv_path_ms_per_step"] = fp32_pass["ms_per_step"]
    if weak is not None:
        line["weak"] = {k: weak[k] for k in ("value", "global_batch", "ms_per_step") if k in weak}
    if os.environ.get("CSEG_BENCH_ROUTE_FALLBACK"):
        line["config"]["route_fallback"] = os.environ["CSEG_BENCH_ROUTE_FALLBACK"][:200]
    detail = {
        "line": dict(line),
        "config": {"workload": wl["name"], "num_classes": cfg.get("data", "num_classes"), "labels": args.labels or wl["labels"],
                   "cross_rank_contrast_set": bool(world > 1 and cfg.exists("contrast", "cross_rank")
                                                   and cfg.get("contrast", "cross_rank")),
                   "conv3x3_arithmetic": conv_arith_note(Kn, split_on), "miopen_find": bool(args.miopen_find),
                   "channels_last": bool(args.channels_last)},
        "roofline": {"traffic_source": traffic_src, "hip_event_ms_per_step": round(step_ms, 3),
                     "roof_ms_per_step": round(roof_s * 1e3, 2), "dominant_kernel": dom,
                     "note": "whole step: images/s (HIP-event time) x %.4f TFLOP/image of fp32-equivalent work; peak = split-operand flops "
                             "at 2500 / %d TFLOP/s + remaining flops at 157.3 TFLOP/s (dense peaks, MI355X_MICROARCH.md)"
                             % (wl["tflop"], MFMAS_PER_PRODUCT[arith])},
        "fp32_conv_path": fp32_pass, "weak": weak, "cpu_baseline": cpu, "kernels": kernels, "split_kernels": split_rows,
        "split_kernel_families": kernel_families(split_rows),
    }
    # the contract line must fit whatever a configuration puts into it: shed the optional fields before a parser loses the head
    for victim in ("weak", "fp32_conv_path_ms_per_step", "detail"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(victim, None)
    if len(json.dumps(line)) > LINE_LIMIT:
        line["config"] = {k: line["config"][k] for k in ("model", "global_batch", "arithmetic", "parallelism")}
        line["config"]["workload"] = args.workload
    return line, detail


def write_detail(detail):
    """bench_detail.json next to bench.py and, when it exists (GPU box: merged back by gpurun), under gpurun_out/."""
    paths = [os.path.join(ROOT, DETAIL_NAME)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", DETAIL_NAME))
    for path in paths:
        try:
            with open(path, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline_worker():
    """CPU port of the train step (model classes of this repo on CPU, device half = oracle/cpu_port.py).
    Runs in its own process (bench.py --cpu-baseline-worker) so that a slow host can be cut off by a timeout.
    Sample: the BASELINE batch of 8 images per step when the host has the memory for it (the reference needs ~34 GB RSS
    at bs8), otherwise 2 images per step; 1 warm-up + 2 timed steps."""
    from oracle import cpu_port
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    cores = usable_cores()
    torch.set_num_threads(cores)
    cpu_port.install(None)
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        avail_gb = 0.0
    batch_size = 8 if (avail_gb >= 96 and cores >= 12) else 2
    cfg = Configer(configs=os.path.join(ROOT, "configs", "cityscapes", "H_48_D_4.json"))
    cfg.update(["train", "batch_size"], batch_size)
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["solver", "max_iters"], 10 ** 9)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    cfg.add(["gpu"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    tr.module_runner.device = lambda: torch.device("cpu")
    tr.seg_net.cpu().train()
    tr.pixel_loss.cpu()
    batch = next(iter(SyntheticLoader(cfg, torch.device("cpu"), length=1, seed=304, mode="uniform")))
    tr.train_step(batch)                       # warm-up
    n = 2                                      # >= 2 timed steps, also at the BASELINE batch of 8 (~42 s each)
    t0 = time.time()
    for _ in range(n):
        tr.train_step(batch)
    dt = (time.time() - t0) / n
    print("CPU_BASELINE " + json.dumps({
        "value": round(batch_size / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
        "sample": "%d images/step (BASELINE bs 8), 1 warm-up + %d timed steps, fp32, %.1f s/step; port pinned to the reference by "
                  "tests/test_step_golden.py::test_sgd_step_cpu_port_matches_reference" % (batch_size, n, dt),
        "host_ram_gb": round(avail_gb)}))


def cpu_baseline(timeout_s=420):
    import subprocess
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], env=env,
                             capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "cpu baseline exceeded %d s and was cut off" % timeout_s, "cores": usable_cores()}
    for line in out.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    return {"error": "cpu baseline failed: " + out.stderr[-400:]}


def timed_steps(tr, batch, steps, warmup, world, device):
    """W untimed steps, then K timed steps bracketed by barrier + synchronize. Returns (wall s, HIP-event ms, loss),
    MAX over ranks."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(warmup):
        tr.train_step(batch)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        loss = tr.train_step(batch)
    e1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt, ev_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
    return dt, ev_ms, float(loss)


# Routes that became defaults in round 4 (forked streams, convolution-epilogue BN statistics, the one-node residual block, hipGraph replay
# at one image per GPU): if a run with them dies or ends with a non-finite loss, the measurement is repeated ONCE with the configuration
# of the round-3 closing tree (one stream, statistics pass, four nodes per block, eager), and the line says so in config.route_fallback.
# Off with CSEG_BENCH_GUARD=0 (e.g. under rocprofv3), and never applied when the caller chose one of these routes explicitly.
SAFE_ROUTES = {"CSEG_BRANCH_STREAMS": "0", "CSEG_CONV_STATS": "0", "CSEG_BLOCK_FUSED": "0", "CSEG_STEP_GRAPH": "0",
               "CSEG_SB16_ROWS8": "0", "CSEG_SB16_XCD": "0"}      # (+ the 4-row tiles and plain tile order of the persistent 3x3 kernels)


def guard_enabled():
    return os.environ.get("CSEG_BENCH_GUARD", "1") != "0" and not any(k in os.environ for k in SAFE_ROUTES)


def run_guarded(cmd_for_attempt):
    """cmd_for_attempt(i) -> argv. Attempt 0 with the defaults; on a non-zero exit, attempt 1 with SAFE_ROUTES."""
    import subprocess
    env = dict(os.environ, CSEG_BENCH_GUARDED="1")
    first = subprocess.run(cmd_for_attempt(0), env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    sys.stdout.write(first.stdout)
    sys.stdout.flush()
    rc = first.returncode
    if rc == 0:
        return 0
    if any(l.startswith("{") and '"metric"' in l for l in first.stdout.splitlines()):
        # the measurement finished and printed its line; only the teardown failed: nothing to repeat
        sys.stderr.write("bench.py: exit code %d AFTER the result line was printed (teardown); keeping that result\n" % rc)
        return 0
    sys.stderr.write("bench.py: the run with the default routes ended with exit code %d; repeating once with %s\n"
                     % (rc, " ".join("%s=%s" % kv for kv in sorted(SAFE_ROUTES.items()))))
    env.update(SAFE_ROUTES)
    env["CSEG_BENCH_ROUTE_FALLBACK"] = ("the first attempt (default routes) ended with exit code %d; this line was measured with "
                                        "%s" % (rc, " ".join("%s=%s" % kv for kv in sorted(SAFE_ROUTES.items()))))
    return subprocess.call(cmd_for_attempt(1), env=env)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) through
    torch.distributed.run on 127.0.0.1 and pass their output through (rank 0 prints the JSON line)."""
    import subprocess
    from contrastiveseg_amd.lib.utils.distributed import respawn_command
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and (args.backend or os.environ.get("CSEG_DIST_BACKEND")) != "gloo":
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible (RCCL needs one device per rank; "
                         "`--backend gloo` lets ranks share a device for a dry run)\n" % (args.gpus, n_dev))
        sys.exit(2)
    make = lambda attempt: respawn_command(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])     # fresh port each time
    if guard_enabled():
        sys.exit(run_guarded(make))
    sys.exit(subprocess.call(make(0), env=dict(os.environ)))


def step_traffic():
    """HBM bytes per step of the default workload at N=1, from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs; gfx950 FETCH_SIZE x2 for wide coalesced reads per MI355X_MICROARCH.md)."""
    for name in ("r06_step_pmc.json", "r05_step_pmc.json", "r04_step_pmc.json", "r03_step_pmc.json", "r02_step_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None, None
    try:
        d = json.load(open(path))
        # (the counter passes carry a stamp of the kernel sources + kernels.py they were taken on, tools/merge_step_pmc.py;
        # a file of another tree gives `traffic: null` and says so instead of passing for a measurement of this tree)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import importlib
        stamp_now = importlib.import_module("merge_step_pmc_stamp").tree_stamp()
        if d.get("tree_stamp") != stamp_now:
            return None, "profiles/%s is stale (stamp %s, this tree %s): no traffic figure" % (name, d.get("tree_stamp"), stamp_now)
        return d["hbm_bytes_per_step"], "profiles/%s, taken on this tree (stamp %s)" % (name, stamp_now)
    except Exception:
        return None, None


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    from contrastiveseg_amd.lib.utils import distributed as D
    if args.gpus > 1 and not D.launched_by_torchrun():
        self_launch(args)
    if args.gpus == 1 and not D.launched_by_torchrun() and not args.in_child and guard_enabled():
        sys.exit(run_guarded(lambda attempt: [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--in-child"]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if args.backend:
        os.environ["CSEG_DIST_BACKEND"] = args.backend
    if args.dist_single_rank:
        assert world == 1, "--dist-single-rank is an N = 1 measurement"
        os.environ.update(CSEG_DIST_SINGLE_RANK="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(D.free_port()))
        torch.cuda.set_device(0)
        torch.distributed.init_process_group(args.backend or "nccl", rank=0, world_size=1)
    D.setup_process_group()
    local = D.device_index()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    from contrastiveseg_amd import kernels as Kn
    if args.conv_arith != "default":
        Kn.CONV3X3_SPLIT_BF16 = args.conv_arith == "split_bf16"
    split_on = bool(Kn.CONV3X3_SPLIT_BF16)

    wl = WORKLOADS[args.workload]
    base_batch = args.global_batch or wl["batch"]
    global_batch = base_batch * world if args.scaling == "weak" else base_batch
    t_start = time.perf_counter()
    tr, cfg, batch = build_trainer(args, world, device, global_batch)
    dt, ev_ms, final_loss = timed_steps(tr, batch, args.steps, args.warmup, world, device)
    if (final_loss != final_loss or abs(final_loss) == float("inf")) and os.environ.get("CSEG_BENCH_GUARDED") == "1" \
            and "CSEG_BENCH_ROUTE_FALLBACK" not in os.environ:
        sys.stderr.write("bench.py: non-finite loss %r with the default routes\n" % final_loss)
        sys.exit(3)                                   # the guarding parent repeats the run with SAFE_ROUTES
    elapsed = torch.tensor([time.perf_counter() - t_start], device=device)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)      # same decision on every rank

    # the same step with the 3x3 convolutions on the pure fp32 path (MIOpen + fp32-MFMA kernel), same trainer, same
    # batch: what the split-bf16 kernels buy, measured under the same clock
    fp32_pass = None
    from contrastiveseg_amd.segmentor.tools import step_graph
    graph_state = os.environ.get("CSEG_STEP_GRAPH_STATE")          # of the TIMED steps (the passes below run eagerly)
    graph_was = step_graph.ENABLED
    step_graph.ENABLED = False            # the comparison pass and the launch tally need the Python path (a replay ignores the switches)
    if split_on and not args.no_fp32_pass:
        Kn.CONV3X3_SPLIT_BF16 = False
        try:                              # never lose the headline number to the comparison pass
            f_steps = max(3, args.steps // 2)
            f_dt, _, _ = timed_steps(tr, batch, f_steps, 2, world, device)
            fp32_pass = {"value": round(global_batch * f_steps / f_dt, 3), "unit": "images/sec", "steps": f_steps,
                         "warmup": 2, "ms_per_step": round(f_dt * 1e3 / f_steps, 3),
                         "conv3x3_arithmetic": "fp32 (MIOpen + fp32-MFMA kernel)"}
        except Exception as e:
            fp32_pass = {"error": repr(e)}
        finally:
            Kn.CONV3X3_SPLIT_BF16 = True

    split_flops = 0.0
    tally = None
    if split_on:
        # one extra untimed step with counting wrappers, on EVERY rank (it contains the step's collectives): which flops run on the
        # split-operand kernels -> the roof of this implementation
        try:
            tally = tally_split_calls(tr, batch, Kn)
            split_flops = split_flops_of(tally)
        except Exception as e:
            sys.stderr.write("bench.py: tally of the split-operand launches failed: %r\n" % (e,))
    step_graph.ENABLED = graph_was
    if graph_state is not None:
        os.environ["CSEG_STEP_GRAPH_STATE"] = graph_state
    weak = None
    # extra weak-scaling pass: RCCL runs only (the gloo dry run shares one GPU between the ranks and says nothing about
    # scaling), and only when the headline measurement itself was quick, so the whole invocation stays within minutes
    if (world > 1 and args.scaling == "strong" and not args.no_weak and torch.distributed.get_backend() == "nccl"
            and float(elapsed) < 180.0):
        # per-GPU work fixed at the config's global batch: separates "the step is too small per GPU" from "the
        # collectives do not overlap". Fewer steps; outside the timed region of the headline number.
        del tr, batch
        torch.cuda.empty_cache()
        w_steps, w_warm = max(3, args.steps // 2), 2
        tr2, _, batch2 = build_trainer(args, world, device, base_batch * world)
        w_dt, _, _ = timed_steps(tr2, batch2, w_steps, w_warm, world, device)
        weak = {"value": round(base_batch * world * w_steps / w_dt, 3), "unit": "images/sec",
                "global_batch": base_batch * world, "per_gpu_batch": base_batch, "steps": w_steps, "warmup": w_warm,
                "ms_per_step": round(w_dt * 1e3 / w_steps, 3)}
        del tr2, batch2
        torch.cuda.empty_cache()

    kernels = None
    split_rows = None
    if rank == 0 and not args.no_kernels and args.workload == "cfg2":
        if world == 1 and split_on and tally is not None:
            try:
                split_rows, split_flops = split_kernel_rooflines(tally, device, Kn)
            except Exception as e:        # never lose the headline number to a micro-benchmark problem
                split_rows = [{"error": repr(e)}]
        tr = None
        torch.cuda.empty_cache()
        try:
            kernels = kernel_rooflines(device, 8)
        except Exception as e:
            kernels = {"error": repr(e)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "cfg2":
        try:
            cpu = cpu_baseline()
        except Exception as e:
            cpu = {"error": repr(e)}
    if world > 1:
        torch.distributed.barrier()

    line = None
    if rank == 0:
        line, detail = assemble_line(args, wl, cfg, world, global_batch, dt, ev_ms, final_loss, split_on, Kn,
                                     torch.distributed.get_backend() if (world > 1 or args.dist_single_rank) else None, fp32_pass, weak, cpu, kernels,
                                     split_rows if (split_rows and "error" not in split_rows[0]) else None, split_flops)
        write_detail(detail)
    if world > 1 or args.dist_single_rank:
        try:                                          # every rank empties its C stdio buffers BEFORE rank 0 prints the line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if line is not None:
        # The contract line must be the LAST line of stdout. RCCL writes its version banner through C stdio, which is block-buffered
        # when stdout is a pipe or a file and would otherwise be flushed at exit -- AFTER a line printed from Python (seen on the MI355X
        # in round 5: `tail -1` of a --dist-single-rank run was "Librccl path : ..."). So: tear the process group down first, flush
        # the C buffers, then print.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stderr.flush()
        sys.stdout.flush()
        print(json.dumps(line), flush=True)           # the LAST line of stdout, <= LINE_LIMIT bytes


if __name__ == "__main__":
    main()
