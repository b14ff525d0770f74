"""Hand-written 3x3 MFMA convolution (csrc/conv3x3.hip, forward + backward-data) against torch's fp64 CPU convolution --
what nn.Conv2d of the reference's BasicBlock computes (lib/models/backbones/hrnet/hrnet_backbone.py:35-66). fp32 MFMA is
an exact FMA chain, so the error is the rounding of a K = 9*Cin term dot product: bar 2e-6 * sqrt(K) of the largest
output. Shapes cover partial tiles (H % 4, W % 64 != 0), Cin != Cout and the module-level dispatch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 48, 48, 9, 12), (1, 96, 48, 7, 68), (2, 48, 96, 5, 132), (1, 48, 48, 128, 256),
                                         (3, 96, 96, 17, 64), (1, 144, 48, 4, 4)])
def test_conv3x3_matches_fp64(B, Ci, Co, H, W):
    dev = _dev()
    from contrastiveseg_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + Ci + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    gy = torch.randn(B, Co, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy.double())
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    assert K.conv3x3_eligible(xd, wd)
    y = K.conv3x3(xd, wd)
    y.backward(gy.to(dev))
    dw_mfma = K._conv3x3_wrw(xd.detach(), gy.to(dev), Co, Ci)        # the MFMA weight-gradient kernel, every shape
    torch.cuda.synchronize()
    tol = 2e-6 * np.sqrt(9 * max(Ci, Co))
    for got, ref in ((y, yr), (xd.grad, xr.grad), (wd.grad, wr.grad)):
        err = float((got.detach().cpu().double() - ref.detach()).abs().max())
        assert err <= tol * max(1.0, float(ref.detach().abs().max())), (err, float(ref.detach().abs().max()))
    # K of the weight gradient = B*H*W pixels
    err = float((dw_mfma.cpu().double() - wr.grad).abs().max())
    assert err <= 2e-6 * np.sqrt(B * H * W) * max(1.0, float(wr.grad.abs().max())), err
    assert torch.equal(dw_mfma, K._conv3x3_wrw(xd.detach(), gy.to(dev), Co, Ci)), "weight gradient not deterministic"


def test_conv3x3_module_dispatch_and_state_dict():
    dev = _dev()
    from contrastiveseg_amd.lib.models.tools.module_helper import Conv3x3
    torch.manual_seed(3)
    m = Conv3x3(48, 48).to(dev)
    torch.manual_seed(3)
    ref = torch.nn.Conv2d(48, 48, 3, 1, 1, bias=False).to(dev)
    assert list(m.state_dict()) == ["weight"] and torch.equal(m.weight, ref.weight)
    x = torch.randn(2, 48, 16, 32, device=dev)
    assert float((m(x) - ref(x)).abs().max()) <= 1e-4             # MFMA kernel vs MIOpen
    x_odd = torch.randn(2, 48, 16, 30, device=dev)                 # W % 4 != 0: since round 5 the split kernel too (element stores)
    assert float((m(x_odd) - ref(x_odd)).abs().max()) <= 1e-4
    m2 = Conv3x3(48, 48, stride=2).to(dev)                         # strided: MIOpen path
    assert m2(x).shape == (2, 48, 8, 16)
