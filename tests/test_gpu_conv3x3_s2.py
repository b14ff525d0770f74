"""3x3 / stride 2 / pad 1 convolutions of HRNet's fuse and transition layers on the split f16x3 kernels (csrc/conv3x3_s2.hip:
forward and backward-data; csrc/conv3x3_sb_wrw.hip: the stride-2 weight gradient) against an fp64 convolution, with MIOpen's fp32
result of the same operands as the yardstick (same bound as tests/test_gpu_conv3x3_sb.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

S2_CASES = [  # B, Cin, Cout, Ho, Wo
    (1, 48, 48, 5, 32),          # ragged row tile, half a column tile, runs of 4 rows in the weight gradient
    (2, 48, 96, 8, 64),          # the shape class of the fuse layers (48 -> 96)
    (1, 96, 192, 6, 96),         # two column tiles, the second ragged; two input-channel blocks in the weight gradient
    (2, 192, 384, 4, 32),        # the widest pair of HRNet-W48
    (1, 256, 96, 8, 64),         # transition 1: 256 input channels = four channel tiles per block in the backward-data kernel
    (2, 64, 64, 8, 64),          # round 6: the second stem convolution (64 -> 64): four tiles per block forward and backward-data, a partly filled channel block in the weight gradient
    (1, 64, 128, 6, 32),
]


def _bound(ref64, got, fp32):
    scale = float(ref64.abs().max())
    err = float((got.double() - ref64).abs().max())
    base = float((fp32.double() - ref64).abs().max())
    return err, max(8.0 * base, 4e-6 * scale)


def _inputs(B, ci, co, Ho, Wo, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, ci, 2 * Ho, 2 * Wo, generator=g).relu_()
    w = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    dy = torch.randn(B, co, Ho, Wo, generator=g) * 1e-3
    return x, w, dy


@pytest.mark.parametrize("case", S2_CASES)
def test_stride2_module_matches_fp64(case, monkeypatch):
    """module_helper.Conv3x3(cin, cout, 2): nn.Conv2d semantics, every direction the kernels cover on them (checked), the rest on
    aten."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import Conv3x3
    B, ci, co, Ho, Wo = case
    x, w, dy = _inputs(*case, seed=5)
    calls = []
    for name in ("conv3x3_s2_run", "conv3x3_s2_bwd_run", "conv3x3_s2_wrw"):
        monkeypatch.setattr(K, name, (lambda fn, name: lambda *a, **k: (calls.append(name), fn(*a, **k))[1])(getattr(K, name), name))
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 2, 1)
    y64.backward(dy.double())
    conv = Conv3x3(ci, co, 2).cuda()
    with torch.no_grad():
        conv.weight.copy_(w.cuda())
    xd = x.cuda().requires_grad_(True)
    y = conv(xd)
    y.backward(dy.cuda())
    xr, wr = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 1)
    yr.backward(dy.cuda())
    want = ["conv3x3_s2_run", "conv3x3_s2_bwd_run"] + (["conv3x3_s2_wrw"] if Wo % 32 == 0 else [])
    assert calls == want, (calls, want)
    for name, ref, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dx", x64.grad, xd.grad, xr.grad),
                                 ("dw", w64.grad, conv.weight.grad, wr.grad)):
        err, tol = _bound(ref, got.cpu(), fp32.cpu())
        assert err <= tol, (case, name, err, tol)


def test_stride2_weight_gradient_is_deterministic():
    from contrastiveseg_amd import kernels as K
    x, w, dy = _inputs(2, 48, 96, 16, 64, seed=6)
    xd, dyd = x.cuda(), dy.cuda()
    a = K.conv3x3_s2_wrw(xd, dyd)
    b = K.conv3x3_s2_wrw(xd, dyd)
    assert torch.equal(a, b)


def test_stride2_at_the_benched_shapes():
    """48 -> 96 at 8 x 128 x 256 -> 64 x 128 (6 channel tiles per block, 16-row runs) and 96 -> 384 at 8 x 32 x 64 -> 16 x 32 (half of
    every column tile is padding), against MIOpen's fp32 result: 2e-5 of the output scale (two fp32 evaluation orders of K = 9 x 96
    products differ by ~1e-6)."""
    from contrastiveseg_amd import kernels as K
    for (B, ci, co, Ho, Wo) in ((8, 48, 96, 64, 128), (8, 96, 384, 16, 32)):
        x, w, dy = _inputs(B, ci, co, Ho, Wo, seed=7)
        xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
        xr, wr = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, 2, 1)
        yr.backward(dyd)
        for name, got, ref in (("y", K.conv3x3_s2_run(xd, wd), yr.detach()), ("dx", K.conv3x3_s2_bwd_run(dyd, wd), xr.grad),
                               ("dw", K.conv3x3_s2_wrw(xd, dyd), wr.grad)):
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (ci, co, name)


def test_transition_256_to_48_module_matches_fp64(monkeypatch):
    """module_helper.Conv3x3(256, 48): the stride-1 transition of HRNet (reference hrnet_backbone.py:635-645) -- forward on the
    16-channel-chunk kernel with streamed weights, backward-data with four channel tiles per block (256 output channels of the
    operator), weight gradient with four 64-wide input-channel blocks; all three checked against fp64."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import Conv3x3
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    B, ci, co, H, W = 2, 256, 48, 12, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, ci, H, W, generator=g).relu_()
    w = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    dy = torch.randn(B, co, H, W, generator=g) * 1e-3
    calls = []
    for name in ("conv3x3_sb_run", "conv3x3_sb_wrw"):
        monkeypatch.setattr(K, name, (lambda fn, name: lambda *a, **k: (calls.append(name), fn(*a, **k))[1])(getattr(K, name), name))
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(dy.double())
    conv = Conv3x3(ci, co).cuda()
    with torch.no_grad():
        conv.weight.copy_(w.cuda())
    xd = x.cuda().requires_grad_(True)
    y = conv(xd)
    y.backward(dy.cuda())
    xr, wr = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(dy.cuda())
    assert calls == ["conv3x3_sb_run", "conv3x3_sb_run", "conv3x3_sb_wrw"], calls
    for name, ref, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dx", x64.grad, xd.grad, xr.grad),
                                 ("dw", w64.grad, conv.weight.grad, wr.grad)):
        err, tol = _bound(ref, got.cpu(), fp32.cpu())
        assert err <= tol, (name, err, tol)


@pytest.mark.parametrize("case", [(2, 64, 128), (1, 18, 260), (8, 512, 1024)])
def test_rgb_stem_module_matches_fp64(case):
    """module_helper.StemConv3x3(64): the first stem convolution on the fp32 kernels of csrc/conv3x3_stem.hip (round 6) against an fp64
    convolution, MIOpen's fp32 result as the yardstick; the weight gradient is bit-reproducible."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import StemConv3x3
    B, H, W = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5.0
    dy = torch.randn(B, 64, H // 2, W // 2, generator=g) * 1e-3
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x.double(), w64, None, 2, 1)
    y64.backward(dy.double())
    conv = StemConv3x3(64).cuda()
    with torch.no_grad():
        conv.weight.copy_(w.cuda())
    xd = x.cuda()
    assert K.conv3x3_s2_rgb_eligible(xd, conv.weight)
    y = conv(xd)
    y.backward(dy.cuda())
    wr = w.cuda().requires_grad_(True)
    yr = F.conv2d(xd, wr, None, 2, 1)
    yr.backward(dy.cuda())
    for name, ref, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dw", w64.grad, conv.weight.grad, wr.grad)):
        err, tol = _bound(ref, got.cpu(), fp32.cpu())
        assert err <= tol, (case, name, err, tol)
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    conv(xd).backward(dy.cuda())
    assert torch.equal(g1, conv.weight.grad), "weight gradient not deterministic"
