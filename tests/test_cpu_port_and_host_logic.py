"""CPU tests of the HOST logic of the product (selection plan + RNG order, loss modules' wiring, enqueue pointer
arithmetic) with the device half replaced by the torch-CPU restatement oracle/cpu_port.py. The port itself is
pinned here against the reference-generated golden vectors, so these tests also check the product's host code
against the reference end to end. The HIP kernels are NOT exercised here (tests/test_gpu_kernels.py does that)."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_port
from oracle.make_golden import ENQ_CASES, LOSS_CASES, case_inputs, enq_init, enq_inputs

SMALL = [n for n, c in LOSS_CASES.items() if c["B"] * c["H"] * c["W"] <= 4 * 128 * 256]


def _configer(c):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    k = dict(proj_dim=c["D"], base_temperature=0.07, use_rmi=False, use_lovasz=False, warmup_iters=0)
    k.update(c["contrast"])
    params = {"ce_ignore_index": -1, "ce_reduction": "elementwise_mean"}
    if c["ce_weight"] is not None:
        params["ce_weight"] = list(c["ce_weight"])
    return Configer(config_dict={"data": {"num_classes": c["K"]},
                                 "network": {"loss_weights": {"aux_loss": 0.4, "seg_loss": 1.0}},
                                 "contrast": k, "loss": {"loss_type": c["loss"], "params": params}})


@pytest.mark.parametrize("name", SMALL)
def test_loss_modules_with_cpu_port_match_reference_golden(name, golden_dir, monkeypatch):
    cpu_port.install(monkeypatch)
    from contrastiveseg_amd.lib.loss.loss_manager import SEG_LOSS_DICT
    c = LOSS_CASES[name]
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % name))
    target, seg, embed, extra = case_inputs(c)
    crit = SEG_LOSS_DICT[c["loss"]](_configer(c))
    t_seg = torch.from_numpy(seg).requires_grad_(True)
    t_embed = torch.from_numpy(embed).requires_grad_(True)
    preds = {"seg": t_seg, "embed": t_embed}
    for k, v in extra.items():
        preds[k] = torch.from_numpy(v)
    torch.manual_seed(c["torch_seed"])
    total = crit(preds, torch.from_numpy(target), with_embed=c.get("with_embed", True))
    total.backward()
    T, V = len(g["anchor_cls"]), int(g["n_view"])
    P = seg.shape[-2] * seg.shape[-1]
    sel = crit.contrast_criterion.last_selection["sel_pix"].numpy().reshape(V, T).T
    assert np.array_equal(sel, g["anchor_img"].astype(np.int64) * P + g["anchor_pix"])
    assert abs(float(total.detach()) - float(g["total"])) < 1e-5 * max(1.0, abs(float(g["total"])))
    assert np.allclose(t_seg.grad.numpy(), g["d_seg"], rtol=1e-3, atol=1e-7)
    ge = t_embed.grad.numpy().reshape(embed.shape[0], embed.shape[1], -1)
    rows = ge[g["anchor_img"].reshape(-1), :, g["anchor_pix"].reshape(-1)]
    assert np.allclose(rows, g["d_embed_rows"], rtol=1e-3, atol=1e-7)


def test_reference_predict_signature_is_accepted(monkeypatch):
    cpu_port.install(monkeypatch)
    from contrastiveseg_amd.lib.loss.loss_contrast import PixelContrastLoss
    c = LOSS_CASES["small_self"]
    target, seg, embed, _ = case_inputs(c)
    crit = PixelContrastLoss(_configer(c))
    t_seg = torch.from_numpy(seg)
    torch.manual_seed(1)
    a = crit(torch.from_numpy(embed), torch.from_numpy(target), predict=torch.max(t_seg, 1)[1])
    torch.manual_seed(1)
    b = crit(torch.from_numpy(embed), torch.from_numpy(target), seg=t_seg)
    assert float(a) == float(b)


def test_keep_rule_and_errors():
    from contrastiveseg_amd.lib.loss.anchor_sampling import NeverTouched, keep_rule, plan_selection
    assert keep_rule(10, 10, 6) == (3, 3)
    assert keep_rule(10, 1, 6) == (5, 1)
    assert keep_rule(1, 10, 6) == (1, 5)
    assert keep_rule(3, 2, 5) == (3, 2)          # n_view/2 = 2.5: hard >= 2.5, easy < 2.5 -> all easy, rest hard
    with pytest.raises(NeverTouched):
        keep_rule(1, 1, 6)
    counts = np.zeros((2, 3, 2), dtype=np.int64)
    assert plan_selection(counts, 64, 5) is None                     # nothing qualifies
    counts[:, :, 0] = 50
    with pytest.raises(RuntimeError):
        plan_selection(counts, 4, 5)                                 # 6 segments > max_samples
    # RNG is consumed even for empty hard/easy sets and in hard-then-easy order
    counts = np.array([[[0, 40], [7, 30]]], dtype=np.int64)
    torch.manual_seed(5)
    plan = plan_selection(counts, 20, 8)
    torch.manual_seed(5)
    torch.randperm(0); e0 = torch.randperm(40)
    h1 = torch.randperm(7); e1 = torch.randperm(30)
    assert plan.n_view == 8 and plan.T == 2
    sel = plan.row_off.reshape(8, 2).T
    assert np.array_equal(sel[0], 0 + e0[:8].numpy())
    assert np.array_equal(sel[1], np.concatenate([40 + h1[:4].numpy(), 47 + e1[:4].numpy()]))


@pytest.mark.parametrize("name", list(ENQ_CASES))
def test_trainer_enqueue_with_cpu_port_matches_reference_golden(name, golden_dir, monkeypatch):
    cpu_port.install(monkeypatch)
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    c = ENQ_CASES[name]
    g = np.load(os.path.join(golden_dir, "%s.npz" % name))
    me = Trainer.__new__(Trainer)
    me.network_stride, me.memory_size, me.pixel_update_freq = c["network_stride"], c["memory_size"], c["pixel_update_freq"]
    sq, pq = enq_init(c)
    sq, pq = torch.from_numpy(sq), torch.from_numpy(pq)
    sp = torch.zeros(c["K"], dtype=torch.long)
    pp = torch.zeros(c["K"], dtype=torch.long)
    torch.manual_seed(c["torch_seed"])
    for r in range(c["rounds"]):
        target, embed = enq_inputs(c, r)
        me._dequeue_and_enqueue(torch.from_numpy(embed), torch.from_numpy(target), sq, sp, pq, pp)
        assert np.array_equal(sp.numpy(), g["segment_ptr_%d" % r])
        assert np.array_equal(pp.numpy(), g["pixel_ptr_%d" % r])
        assert np.allclose(sq.numpy(), g["segment_queue_%d" % r], rtol=1e-5, atol=1e-6)
        assert np.allclose(pq.numpy(), g["pixel_queue_%d" % r], rtol=1e-5, atol=1e-6)


def test_config1_plumbing_trainer_steps_on_cpu(monkeypatch):
    """BASELINE.json configs[0]: ResNet-18 DeepLab-V3, 4x 256x256x5-class, 128-d projection, 256 anchors, CPU.
    Trainer -> registry -> loss -> backward -> SGD; loss decreases on a fixed batch."""
    cpu_port.install(monkeypatch)
    import os as _os
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    cfg = Configer(configs=_os.path.join(root, "configs", "synthetic", "R_18_D_8_tiny.json"))
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    cfg.add(["gpu"], None)
    cfg.get("train", "data_transformer")["input_size"] = [96, 96]
    cfg.update(["solver", "max_iters"], 3)
    cfg.update(["contrast", "max_views"], 10)
    torch.manual_seed(304)
    tr = Trainer.__new__(Trainer)
    loader = None
    Trainer.__init__(tr, cfg, train_loader=loader)
    tr.train_loader = SyntheticLoader(cfg, torch.device("cpu"), length=3, mode="blocky")
    tr.seg_net.train()
    losses = [float(tr.train_step(b)) for b in tr.train_loader]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert cfg.get("iters") == 3


@pytest.mark.parametrize("model,backbone,cfg_file", [
    ("deeplab_v3_mem", "deepbase_resnet18_dilated8", "cityscapes/R_101_D_8_MEM.json"),
    ("hrnet_w48_ocr_mem", "hrnet18", "coco_stuff/H_48_D_4_MEM.json")])
def test_memory_variants_of_baseline_configs_3_and_4_step_on_cpu(model, backbone, cfg_file, monkeypatch):
    """deeplab_v3_mem / hrnet_w48_ocr_mem + mem_contrast_auxce_loss (the buildable forms of BASELINE.json configs[3]/[4],
    SURVEY.md section 7): registry, forward(img, labels) contract, aux CE + bank contrast, enqueue, SGD -- host logic
    with the device half replaced by oracle/cpu_port.py."""
    cpu_port.install(monkeypatch)
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Configer(configs=os.path.join(root, "configs", cfg_file))
    assert cfg.get("network", "model_name") == model and cfg.get("loss", "loss_type") == "mem_contrast_auxce_loss"
    cfg.update(["network", "backbone"], backbone)
    cfg.update(["network", "bn_type"], "torchbn")
    cfg.update(["data", "num_classes"], 6)
    cfg.get("loss", "params").pop("ce_weight", None)
    cfg.update(["train", "batch_size"], 2)
    cfg.get("train", "data_transformer")["input_size"] = [96, 64]
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["contrast", "memory_size"], 16)
    cfg.update(["solver", "max_iters"], 2)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    cfg.add(["gpu"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    loader = SyntheticLoader(cfg, torch.device("cpu"), length=2, mode="blocky")
    tr.seg_net.train()
    w0 = next(tr.seg_net.parameters()).detach().clone()
    losses = [float(tr.train_step(b)) for b in loader]
    assert all(np.isfinite(losses)), losses
    assert not torch.equal(w0, next(tr.seg_net.parameters()).detach())
    assert int(tr.seg_net.pixel_queue_ptr.sum()) > 0
    assert set(k for k in tr.seg_net.state_dict() if "queue" in k) == {"segment_queue", "segment_queue_ptr",
                                                                       "pixel_queue", "pixel_queue_ptr"}
