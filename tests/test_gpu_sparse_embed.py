"""Row-sparse backward of the projection head on the GPU (HIP BN kernels + rocBLAS / MIOpen + the HIP contrast
kernels) against the dense route, at the head's real width. Behind CSEG_TEST_SPARSE_EMBED=1 until it has run on hardware
once (written after the round's GPU budget was spent; the CPU twin is tests/test_sparse_embed_grad.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu          # the row-sparse route is the default since round 3


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _run(sparse, loss_type, monkeypatch):
    from contrastiveseg_amd import kernels as Kn
    from contrastiveseg_amd.lib.loss.loss_manager import SEG_LOSS_DICT
    from contrastiveseg_amd.lib.models.modules.projection import ProjectionHead
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from oracle import cseg_oracle as O
    monkeypatch.setattr(Kn, "SPARSE_EMBED_GRAD", sparse)
    K, D, C = 19, 256, 720
    dev = _dev()
    target, seg, _ = O.synth_case(5, 2, K, 128, 256, 4, 8)
    torch.manual_seed(11)
    head = ProjectionHead(C, D, bn_type='torchbn').to(dev).train()
    feats = torch.randn(2, C, 32, 64, generator=torch.Generator().manual_seed(3)).to(dev).requires_grad_(True)
    cfg = Configer(config_dict={
        "data": {"num_classes": K}, "network": {"loss_weights": {"aux_loss": 0.4, "seg_loss": 1.0}},
        "contrast": {"proj_dim": D, "temperature": 0.1, "base_temperature": 0.07, "max_samples": 1024, "max_views": 100,
                     "loss_weight": 0.1, "use_rmi": False, "memory_size": 32},
        "loss": {"loss_type": loss_type, "params": {"ce_ignore_index": -1, "ce_reduction": "elementwise_mean"}}})
    crit = SEG_LOSS_DICT[loss_type](cfg).to(dev)
    embed = head(feats)
    preds = {"seg": torch.from_numpy(seg).to(dev).requires_grad_(True), "embed": embed}
    if loss_type.startswith("mem"):
        g = torch.Generator().manual_seed(9)
        for name in ("segment_queue", "pixel_queue"):
            preds[name] = torch.nn.functional.normalize(torch.randn(K, 32, D, generator=g), dim=2).to(dev)
    torch.manual_seed(304)
    loss = crit(preds, torch.from_numpy(target).to(dev), with_embed=True)
    loss.backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    out = {"loss": loss.detach().cpu(), "embed": embed.detach().cpu(), "d_feats": feats.grad.cpu()}
    out.update({"d_" + n: p.grad.cpu() for n, p in head.named_parameters()})
    return out


@pytest.mark.parametrize("loss_type", ["contrast_ce_loss", "mem_contrast_ce_loss"])
def test_sparse_route_equals_dense_route_on_the_gpu(loss_type, monkeypatch):
    dense = _run(False, loss_type, monkeypatch)
    sparse = _run(True, loss_type, monkeypatch)
    assert torch.equal(dense["embed"], sparse["embed"])
    assert float(dense["loss"]) == float(sparse["loss"])
    grads = [k for k in dense if k.startswith("d_")]
    floor = 1e-3 * max(float(dense[k].abs().max()) for k in grads)
    for k in grads:
        ref = dense[k]
        err = float((sparse[k] - ref).abs().max())
        if k == "d_proj.0.bias":            # exactly-zero gradient in front of a training-mode BN: noise on both routes
            assert max(err, float(ref.abs().max())) <= 1e-2 * floor, (k, err)
            continue
        assert err <= 1e-4 * max(float(ref.abs().max()), floor), (k, err, float(ref.abs().max()))
