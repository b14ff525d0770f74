"""Round 6: a forked step is BIT-identical to the single-stream step since HighResolutionModule._exchange_forked creates its autograd
nodes in the single-stream order (before, outputs 1.. were built before output 0, backward summed the gradients that meet at a branch
output in another order, and one SGD step amplified the last-bit difference to 1e-3 of the next loss: profiles/r06_stream_bisect.txt).

Concurrency inside the eager step (round 4): the HRNet branches / exchange paths on forked HIP streams
(lib/models/backbones/hrnet_backbone.py) and the weight gradients on their own stream (kernels.wgrad_scope, opened by
Trainer.train_step). Both only re-order independent work, so one train step must give the gradients of the single-stream step:
compared after the FIRST backward (same weights, same input; later steps of these freshly initialised networks amplify rounding
differences chaotically, see tests/test_gpu_step_graph.py), with MIOpen's deterministic solvers and the single-stream path measured
against itself first."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(forks, wgrad, monkeypatch, model="hrnet_w48_contrast", backbone="hrnet48", batch=2, size=(256, 128)):
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools import step_graph
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    monkeypatch.setattr(HB, "EAGER_FORKS", forks)
    monkeypatch.setattr(HB, "EAGER_FORK_MIN_PIXELS", 0)          # (the default keeps small maps like this test's on one stream)
    monkeypatch.setattr(K, "WGRAD_STREAM", wgrad)
    monkeypatch.setattr(step_graph, "ENABLED", False)
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV1X1_SB_MIN_TILES", 1)
    cfg = Configer(configs=os.path.join(ROOT, "configs", "cityscapes", "H_48_D_4.json"))
    cfg.update(["network", "backbone"], backbone)
    cfg.update(["network", "model_name"], model)
    cfg.update(["data", "num_classes"], 7)
    cfg.get("loss", "params").pop("ce_weight", None)
    cfg.update(["train", "batch_size"], batch)
    cfg.get("train", "data_transformer")["input_size"] = list(size)
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["contrast", "max_views"], 12)
    cfg.update(["solver", "max_iters"], 1000)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    for m in tr.seg_net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    data = list(SyntheticLoader(cfg, tr.module_runner.device(), length=1, mode="blocky", fixed=True))[0]
    tr.seg_net.train()
    tr.pixel_loss.train()
    torch.manual_seed(17)
    l0 = float(tr.train_step(data))
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in tr.seg_net.named_parameters() if p.grad is not None}
    l1 = float(tr.train_step(data))
    torch.cuda.synchronize()
    crit = getattr(tr.pixel_loss, "module", tr.pixel_loss)
    terms = tuple(float(t) for t in crit.last_terms)                      # (segmentation term, contrastive term) of the second step
    sel = crit.contrast_criterion.last_selection["sel_pix"].cpu().numpy().copy()
    del tr, data
    torch.cuda.empty_cache()
    _run.last_second_step = (terms, sel)
    return (l0, l1), grads


def test_forked_streams_and_wgrad_stream_give_the_single_stream_gradients(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    base, g0 = _run(False, False, monkeypatch)
    again, g0b = _run(False, False, monkeypatch)
    report = []
    for name, forks, wgrad in (("forks", True, False), ("wgrad", False, True), ("forks+wgrad", True, True)):
        got, g = _run(forks, wgrad, monkeypatch)
        assert abs(got[0] - base[0]) <= 2e-6 * abs(base[0]), (name, got, base)
        assert set(g) == set(g0)
        gnorm = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g0.values()))
        worst, bad = 0.0, []
        for k, a in g0.items():
            den = max(float(np.linalg.norm(a)), 1e-6 * gnorm)
            dev, own = float(np.linalg.norm(a - g[k])) / den, float(np.linalg.norm(a - g0b[k])) / den
            worst = max(worst, dev)
            if dev > max(4.0 * own, 2e-4):
                bad.append((k, "%.2e" % dev, "%.2e" % own))
        assert not bad, (name, len(bad), bad[:12], got, base, again)
        assert abs(got[1] - base[1]) <= max(2e-5 * abs(base[1]), 4 * abs(again[1] - base[1])), (name, got, base, again, worst)
        report.append("%s: losses %s, worst gradient deviation %.1e" % (name, [round(v, 6) for v in got], worst))
    print("single stream:", [round(v, 6) for v in base], "|", "; ".join(report))
