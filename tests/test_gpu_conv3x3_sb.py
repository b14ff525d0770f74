"""Split-bf16 ("bf16x6") 3x3 convolution on the BF16 matrix cores (csrc/conv3x3_sb.hip) against an fp64 convolution,
with MIOpen's fp32 convolution of the same operands as the yardstick for "fp32 rounding class"."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W
    (1, 48, 48, 5, 8),          # one full 32-channel chunk + the 16-channel tail, one partial tile
    (2, 48, 48, 9, 68),         # ragged tiles in both directions
    (2, 96, 96, 16, 64),        # NT = 6
    (1, 144, 144, 4, 64),       # NT = 9, tail chunk
    (1, 720, 720, 8, 64),       # the head's channel count (22 full chunks + tail, 5 channel tiles of 144)
    (1, 192, 48, 7, 36),        # Cin != Cout
    # round 5: widths that are not multiples of 4 floats (rows not 16-byte aligned: element stores, cseg_store_row4)
    (2, 48, 48, 9, 65),         # 1/8-resolution maps of HRNet at 520 x 520; two column tiles
    (1, 96, 96, 6, 33),
    (1, 192, 192, 4, 130),      # W % 4 == 2
    (1, 144, 144, 5, 129),      # NT = 9 at the width of DeepLab-R101-d8's maps
]


def _inputs(B, ci, co, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    b = torch.randn(co, generator=g)
    return x, w, b


def _bound(ref64, got, fp32):
    """fp32 rounding class: within 8x the deviation of MIOpen's fp32 convolution from the fp64 result (measured on
    MI355X: 2-5x, e.g. 1.7e-5 vs 6.5e-6 at K = 720*9 for outputs up to 5.5 -- the three dropped piece products are each
    2^-24 of the leading one, the same order as one fp32 rounding), floor 4e-6 of the output scale. For comparison, bf16
    operands without splitting sit at 1e-2 and TF32 at 1e-3 of the scale."""
    scale = float(ref64.abs().max())
    err = float((got.double() - ref64).abs().max())
    base = float((fp32.double() - ref64).abs().max())
    return err, max(8.0 * base, 4e-6 * scale)


@pytest.mark.parametrize("glds", ["1", "0"])
@pytest.mark.parametrize("case", CASES)
def test_forward_matches_fp64(case, glds, monkeypatch):
    from contrastiveseg_amd import kernels as K
    monkeypatch.setenv("CSEG_CONV3X3_SB_GLDS", glds)
    B, ci, co, H, W = case
    x, w, b = _inputs(*case)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    assert K.conv3x3_sb_eligible(xd, wd)
    got = K.conv3x3_sb_run(xd, wd, False, bd).cpu()
    fp32 = F.conv2d(xd, wd, bd, 1, 1).cpu()
    err, tol = _bound(ref, got, fp32)
    assert err <= tol, (case, glds, err, tol)
    got_nb = K.conv3x3_sb_run(xd, wd, False, None).cpu()
    assert float((got_nb.double() - (ref - b.double().view(1, -1, 1, 1))).abs().max()) <= tol


@pytest.mark.parametrize("case", CASES)
def test_autograd_matches_fp64(case, monkeypatch):
    """Forward / backward-data on the split-bf16 kernel, weight and bias gradient on MIOpen (the split-bf16 weight-gradient
    route through autograd has its own file, tests/test_zz_gpu_default_routes.py, which runs last)."""
    from contrastiveseg_amd import kernels as K
    monkeypatch.setattr(K, "CONV3X3_SB_WRW", False)
    B, ci, co, H, W = case
    x, w, b = _inputs(*case, seed=1)
    dy = torch.randn(B, co, H, W, generator=torch.Generator().manual_seed(2))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    F.conv2d(x64, w64, b64, 1, 1).backward(dy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    K.conv3x3_split_bf16(xd, wd, bd).backward(dy.cuda())
    xr, wr, br = (t.cuda().requires_grad_(True) for t in (x, w, b))
    F.conv2d(xr, wr, br, 1, 1).backward(dy.cuda())
    for name, g64, got, fp32 in (("dx", x64.grad, xd.grad, xr.grad), ("dw", w64.grad, wd.grad, wr.grad),
                                 ("db", b64.grad, bd.grad, br.grad)):
        err, tol = _bound(g64, got.cpu(), fp32.cpu())
        assert err <= tol, (case, name, err, tol)


# (2, 48, 48, 40, 128): several runs and segments per split -- add once it has run on hardware
WRW_CASES = [  # B, Cin, Cout, H, W
    (1, 48, 48, 5, 64),         # ragged 64-wide channel block (3 of 4 tiles), one run
    (2, 16, 48, 9, 128),        # two column segments, two runs (8 + 1 rows)
    (1, 80, 96, 3, 64),         # two channel blocks each way
    (2, 96, 96, 16, 64),
    (1, 720, 720, 8, 64),       # the head's channel count
    # round 5 (version 2, f16x3 only): widths that are not multiples of the row segment -- the ragged loaders
    (2, 48, 48, 5, 65),         # 64 + 1
    (1, 64, 96, 9, 33),         # 32 + 1, two channel blocks on the output side
    (1, 16, 48, 18, 130),       # 2 x 64 + 2, two runs
    (2, 96, 48, 3, 17),         # narrower than a segment
    # round 6 (version 2, f16x3 only): output channels % 16 but not % 48 -- a partly filled last channel block
    (2, 64, 64, 16, 64),        # the 3x3 convolutions of HRNet's layer 1
    (1, 32, 128, 9, 96),
    (1, 64, 256, 5, 65),        # with a ragged width
]


@pytest.mark.parametrize("version", ["1", "2"])
@pytest.mark.parametrize("case", WRW_CASES)
def test_weight_gradient_matches_fp64(case, version, monkeypatch):
    from contrastiveseg_amd import kernels as K
    monkeypatch.setenv("CSEG_CONV3X3_SB_WRW_V", version)
    B, ci, co, H, W = case
    if (W % 32 or co % 48) and version == "1":
        pytest.skip("version 1 has no ragged loaders / no partly filled channel blocks")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, ci, H, W, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    w64 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(dy.double())
    xd, dyd = x.cuda(), dy.cuda()
    assert K.conv3x3_sb_wrw_eligible(xd, dyd)
    got = K.conv3x3_sb_wrw(xd, dyd)
    wr = torch.zeros(co, ci, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(xd, wr, None, 1, 1).backward(dyd)
    err, tol = _bound(w64.grad, got.cpu(), wr.grad.cpu())
    assert err <= tol, (case, err, tol)
    assert torch.equal(got, K.conv3x3_sb_wrw(xd, dyd)), "weight gradient not deterministic"


@pytest.mark.parametrize("nt", [3, 6])
def test_explicit_channel_tiling_matches_default(nt):
    """cseg_conv3x3_sb_*_nt: same convolution whatever the number of channel tiles per block."""
    from contrastiveseg_amd import kernels as K
    x, w, b = _inputs(2, 96, 96, 9, 68, seed=5)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    got = K.conv3x3_sb_run(xd, wd, False, bd, nt=nt).cpu()
    err, tol = _bound(ref, got, F.conv2d(xd, wd, bd, 1, 1).cpu())
    assert err <= tol, (nt, err, tol)
    dx = K.conv3x3_sb_run(xd, wd, True, None, nt=nt).cpu()
    assert float((dx - K.conv3x3_sb_run(xd, wd, True).cpu()).abs().max()) <= 1e-5


ONE_CASES = [  # B, Cin, Cout, H, W
    (1, 48, 64, 10, 30),        # ragged pixel tile (300 pixels), 16-channel tail
    (2, 144, 48, 16, 16),
    (1, 720, 720, 8, 64),       # projection head, first layer
    (1, 720, 256, 8, 64),       # projection head, second layer (NT = 8)
    (2, 64, 256, 12, 20),       # layer1 bottleneck expansion
    (2, 64, 256, 5, 13),        # round 5: planes that are not multiples of 4 / 32 pixels (65, 559, 8385 = 65 x 129)
    (1, 48, 144, 13, 43),
    (1, 256, 64, 65, 129),
    (1, 64, 512, 8, 24),        # round 5: 16 channel tiles per block in f16x3 (ASPP / OCR widths)
    (1, 48, 240, 5, 13),        # ... and 15 (the 720-channel head's tiling) on a ragged plane
]


@pytest.mark.parametrize("case", ONE_CASES)
def test_pointwise_matches_fp64(case):
    from contrastiveseg_amd import kernels as K
    B, ci, co, H, W = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5
    b = torch.randn(co, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    y64 = F.conv2d(x64, w64, b64)
    y64.backward(dy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    assert K.conv1x1_sb_eligible(xd, wd)
    y = K.conv1x1_split_bf16(xd, wd, bd)
    y.backward(dy.cuda())
    xr, wr, br = (t.cuda().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br)
    yr.backward(dy.cuda())
    for name, t64, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dx", x64.grad, xd.grad, xr.grad),
                                 ("dw", w64.grad, wd.grad, wr.grad), ("db", b64.grad, bd.grad, br.grad)):
        err, tol = _bound(t64, got.cpu(), fp32.cpu())
        assert err <= tol, (case, name, err, tol)


@pytest.mark.parametrize("case", [(2, 48, 64, 8, 8), (1, 144, 160, 8, 12), (1, 720, 720, 8, 64), (1, 720, 256, 8, 64),
                                  (2, 64, 256, 16, 16), (2, 48, 64, 5, 13), (1, 144, 160, 13, 43), (1, 256, 64, 65, 129)])
def test_pointwise_weight_gradient_matches_fp64(case):
    from contrastiveseg_amd import kernels as K
    B, ci, co, H, W = case
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, ci, H, W, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    ref = torch.einsum("bohw,bchw->oc", dy.double(), x.double()).reshape(co, ci, 1, 1)
    xd, dyd = x.cuda(), dy.cuda()
    assert K.conv1x1_sb_wrw_eligible(xd, dyd)
    got = K.conv1x1_sb_wrw(xd, dyd)
    fp32 = torch.einsum("bohw,bchw->oc", dyd, xd).reshape(co, ci, 1, 1)
    err, tol = _bound(ref, got.cpu(), fp32.cpu())
    assert err <= tol, (case, err, tol)
    assert torch.equal(got, K.conv1x1_sb_wrw(xd, dyd)), "weight gradient not deterministic"


@pytest.mark.parametrize("case", [(1, 48, 144, 9, 68), (2, 720, 720, 8, 64)])
def test_head_kernel_8_rows_matches_fp64(case):
    """nt = CSEG_NT_SB8: the 8 x 64-pixel kernel the 720 -> 720 head takes at the benched batch (csrc/conv3x3_sb16.hip, namespace
    sb8), forward with bias and the backward-data operator, against fp64 with MIOpen's fp32 result as the yardstick."""
    from contrastiveseg_amd import kernels as K
    B, ci, co, H, W = case
    x, w, b = _inputs(*case, seed=11)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    got = K.conv3x3_sb_run(xd, wd, False, bd, K.NT_SB8).cpu()
    err, tol = _bound(ref, got, F.conv2d(xd, wd, bd, 1, 1).cpu())
    assert err <= tol, (case, err, tol)
    if ci % 144 == 0:
        dy = torch.randn(B, co, H, W, generator=torch.Generator().manual_seed(12))
        x64 = x.double().requires_grad_(True)
        F.conv2d(x64, w.double(), None, 1, 1).backward(dy.double())
        xr = xd.clone().requires_grad_(True)
        F.conv2d(xr, wd, None, 1, 1).backward(dy.cuda())
        got = K.conv3x3_sb_run(dy.cuda(), wd, True, None, K.NT_SB8).cpu()
        err, tol = _bound(x64.grad, got, xr.grad.cpu())
        assert err <= tol, (case, "dx", err, tol)


@pytest.mark.parametrize("case", [(2, 48, 8, 64), (1, 96, 6, 36)])
def test_basic_block_with_fused_identity_gradient(case, monkeypatch):
    """hrnet_backbone.BasicBlock: the first convolution and the identity path leave ONE autograd node (kernels.Conv3x3SplitFork) whose
    backward adds the identity path's gradient in the epilogue of the backward-data kernel. Output, input gradient and both weight
    gradients against the same block in fp64 (torch), and against the un-fused route of this package."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "BLOCK_FUSED", False)        # this test is about the two-node route; the one-node block is compared below
    B, C, H, W = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    torch.manual_seed(5)
    blk = BasicBlock(C, C, bn_type="torchbn").cuda().train()
    calls = []
    monkeypatch.setattr(K, "conv3x3_split_fork", (lambda fn: lambda *a, **k: (calls.append("fork"), fn(*a, **k))[1])(K.conv3x3_split_fork))

    def run(fork):
        monkeypatch.setattr(K, "CONV3X3_FORK", fork)
        blk.zero_grad(set_to_none=True)
        xd = x.clone().cuda().requires_grad_(True)
        y = blk(xd * 1.0)                 # (a non-leaf input, as inside the network)
        y.backward(dy.cuda())
        return y.detach().cpu(), xd.grad.cpu(), blk.conv1.weight.grad.cpu().clone(), blk.conv2.weight.grad.cpu().clone()

    got = run(True)
    assert calls == ["fork"], calls
    plain = run(False)
    assert calls == ["fork"]
    # the whole block as ONE autograd node (kernels.BasicBlockSplit, round 4): the same kernels in the same order
    monkeypatch.setattr(K, "BLOCK_FUSED", True)
    one = run(True)
    monkeypatch.setattr(K, "BLOCK_FUSED", False)
    if W % 32 == 0:                                     # (narrower maps keep the two-node route: no split weight gradient there)
        assert calls == ["fork"], "the fused block must not go through the fork node"
        for a, b in zip(got, one):
            assert torch.equal(a, b)
    # fp64 reference of the same block
    w1, w2 = blk.conv1.weight.detach().cpu().double().requires_grad_(True), blk.conv2.weight.detach().cpu().double().requires_grad_(True)
    g1, b1 = blk.bn1.weight.detach().cpu().double(), blk.bn1.bias.detach().cpu().double()
    g2, b2 = blk.bn2.weight.detach().cpu().double(), blk.bn2.bias.detach().cpu().double()
    x64 = x.detach().double().requires_grad_(True)
    o = F.relu(F.batch_norm(F.conv2d(x64, w1, None, 1, 1), None, None, g1, b1, True, 0.1, 1e-5))
    o = F.relu(F.batch_norm(F.conv2d(o, w2, None, 1, 1), None, None, g2, b2, True, 0.1, 1e-5) + x64)
    o.backward(dy.double())
    for name, ref, a, p in (("y", o.detach(), got[0], plain[0]), ("dx", x64.grad, got[1], plain[1]), ("dw1", w1.grad, got[2], plain[2]),
                            ("dw2", w2.grad, got[3], plain[3])):
        scale = float(ref.abs().max())
        assert float((a.double() - ref).abs().max()) <= 2e-4 * scale, (case, name)
        assert float((a - p).abs().max()) <= 2e-5 * scale, (case, name, "fused vs plain")


def test_eight_row_tiles_on_hardware(monkeypatch):
    """conv3x3_sb16r_kernel (8 x 64-pixel tiles, one wave per output row, default where the tiles fill 256 blocks -- the 48-channel
    branches at the benched batch): forward with bias, the BatchNorm statistics of its epilogue, and backward-data with the residual
    addend, against the 4-row kernels on the same operands (bit-identical: same packed weights, same accumulation order) and against
    fp64. 6 x 48 x 68 x 264: 270 tiles of 8 rows, ragged in both directions."""
    from contrastiveseg_amd import kernels as K
    B, C, H, W = 6, 48, 68, 264
    x, w, b = _inputs(B, C, C, H, W, seed=7)
    x = x.relu()
    addend = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(8))
    xd, wd, bd, ad = x.cuda(), w.cuda(), b.cuda(), addend.cuda()

    def run(rows8):
        monkeypatch.setenv("CSEG_SB16_ROWS8", rows8)
        y = K.conv3x3_sb_run(xd, wd, False, bd, want_stats=True)
        st = K.known_tile_stats(y)
        dx = K.conv3x3_sb_run(xd, wd, True, None, addend=ad)           # backward-data operator (transposed, flipped) + epilogue addend
        return y, (None if st is None else st.clone()), dx

    y8, st8, dx8 = run("1")
    y4, st4, dx4 = run("0")
    assert torch.equal(y8, y4) and torch.equal(dx8, dx4), "8-row and 4-row kernels must agree bit for bit"
    if st8 is not None and st4 is not None:
        assert st8.shape == st4.shape and float((st8 - st4).abs().max()) <= 1e-5 * float(st4.abs().max())
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    fp32 = F.conv2d(xd, wd, bd, 1, 1).cpu()
    err, tol = _bound(ref, y8.cpu(), fp32)
    assert err <= tol, (err, tol)
    ref_dx = F.conv_transpose2d(x.double(), w.double(), None, 1, 1) + addend.double()
    fp32_dx = (F.conv_transpose2d(xd, wd, None, 1, 1) + ad).cpu()
    err, tol = _bound(ref_dx, dx8.cpu(), fp32_dx)
    assert err <= tol, ("backward-data", err, tol)


@pytest.mark.parametrize("case", [(1, 64, 64, 7, 65, 2), (2, 256, 256, 9, 129, 2), (1, 512, 512, 6, 129, 4), (2, 128, 64, 5, 36, 4)])
def test_dilated_convolution_matches_fp64(case):
    """Round 5: cseg_conv3x3_split_dil_fwd (rate 2 / 4, padding = rate; layer3 / layer4 of DeepLab-V3's dilated ResNet, reference
    lib/models/backbones/resnet/resnet_backbone.py:88-101) -- forward with bias, backward-data and the autograd wrapper (weight / bias
    gradient on MIOpen) against float64, MIOpen's fp32 dilated convolution as the yardstick for "fp32 rounding class"."""
    from contrastiveseg_amd import kernels as K
    B, ci, co, H, W, d = case
    g = torch.Generator().manual_seed(41 + W)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    b = torch.randn(co, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    y64 = F.conv2d(x64, w64, b64, 1, d, d)
    y64.backward(dy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    assert K.conv3x3_dil_eligible(xd, wd, (d, d))
    y = K.conv3x3_dil_split(xd, wd, bd, d)
    y.backward(dy.cuda())
    xr, wr, br = (t.cuda().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, 1, d, d)
    yr.backward(dy.cuda())
    for name, t64, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dx", x64.grad, xd.grad, xr.grad),
                                 ("dw", w64.grad, wd.grad, wr.grad), ("db", b64.grad, bd.grad, br.grad)):
        err, tol = _bound(t64, got.cpu(), fp32.cpu())
        assert err <= tol, (case, name, err, tol)
