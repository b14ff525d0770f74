"""The classifier convolution (csrc/cls1x1.hip, kernels.cls1x1): 1x1 onto K <= 32 classes with the Dropout2d in front of it folded into
per-image weights -- module_helper.FoldedDropout2d + ClassifierConv1x1 against the reference's nn.Dropout2d + nn.Conv2d
(lib/models/nets/hrnet.py:73-80) on the same device with the same generator state: the masks must be the SAME draws, outputs and
all gradients equal within fp32 summation-order noise (fp64 as the yardstick). Replayed on the CPU emulation by tests/test_emu_cabi.py."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


CASES = [  # B, C, K, H, W, bias, dropout p
    (2, 96, 19, 6, 20, False, 0.10),       # the head's structure: no bias, dropout, 19 classes (KP = 20); 120 pixels: a ragged forward tile
    (1, 720, 19, 4, 36, False, 0.10),      # the head's channel count: 11 full channel tiles + 16 channels, odd channel halves
    (3, 64, 7, 5, 13, True, 0.0),          # bias, no dropout, 65 pixels (not a multiple of 4: the scalar loader of the weight gradient)
    (2, 40, 32, 8, 32, True, 0.25),        # 32 classes (KP = 32), 256 pixels
    (1, 512, 21, 3, 24, True, 0.0),        # the OCR classifier's channel count, 21 classes (KP = 32 with pad columns)
]


@pytest.mark.parametrize("case", CASES)
def test_classifier_with_folded_dropout_matches_the_reference_modules(case, monkeypatch):
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import ClassifierConv1x1, FoldedDropout2d
    B, C, Kc, H, W, bias, p = case
    dev = _dev()
    g = torch.Generator().manual_seed(100 + C + Kc)
    x0 = torch.randn(B, C, H, W, generator=g).relu_()
    w0 = torch.randn(Kc, C, 1, 1, generator=g) / C ** 0.5
    b0 = torch.randn(Kc, generator=g) if bias else None
    dy0 = torch.randn(B, Kc, H, W, generator=g)

    def run(fast, dtype):
        conv = ClassifierConv1x1(C, Kc, kernel_size=1, bias=bias) if fast else nn.Conv2d(C, Kc, kernel_size=1, bias=bias)
        drop = FoldedDropout2d(p, conv) if fast else nn.Dropout2d(p)
        if dtype == torch.float64:
            # the fp64 yardstick multiplies by the mask the fp32 runs draw (bernoulli_ on a double tensor is another stream): F.dropout2d
            # of a [B, C, 1, 1] tensor of ones makes the same draws as of the activation (the noise has that shape either way)
            torch.manual_seed(77)
            m = nn.functional.dropout2d(torch.ones(B, C, 1, 1, device=dev), p, True).double()
            drop = type("Mask", (nn.Module,), {"forward": lambda self, t: t * m})()
        net = nn.Sequential(drop, conv).to(dev).to(dtype)
        with torch.no_grad():
            conv.weight.copy_(w0.to(dtype))
            if bias:
                conv.bias.copy_(b0.to(dtype))
        net.train()
        x = x0.to(dev).to(dtype).clone().requires_grad_(True)           # (a fresh leaf: on the emulated device .to() is the identity)
        torch.manual_seed(77)
        calls = []
        if fast:
            orig = K.Cls1x1.apply
            monkeypatch.setattr(K.Cls1x1, "apply", staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
        y = net(x)
        if fast:
            assert calls, "the classifier did not take the cls1x1 kernels"
        after = torch.rand(4, device=dev)                           # the generator must be where the reference leaves it
        y.backward(dy0.to(dev).to(dtype))
        return [t.detach().double().cpu() for t in (y, x.grad, conv.weight.grad) + ((conv.bias.grad,) if bias else ())] + [after.double().cpu()]

    ref64 = run(False, torch.float64)
    ref32 = run(False, torch.float32)
    got = run(True, torch.float32)
    assert torch.equal(got[-1], ref32[-1]), "the folded dropout consumed other generator draws than nn.Dropout2d"
    for name, a, r32, r64 in zip(("y", "dx", "dw", "db"), got[:-1], ref32[:-1], ref64[:-1]):
        scale = float(r64.abs().max())
        err, base = float((a - r64).abs().max()), float((r32 - r64).abs().max())
        assert err <= max(8.0 * base, 4e-6 * scale), (name, err, base, scale)
    # zeroed channels of the mask get exactly zero gradient, as under nn.Dropout2d
    if p > 0:
        zero_ref = (ref32[1].abs().amax((2, 3)) == 0)
        assert torch.equal(got[1].abs().amax((2, 3)) == 0, zero_ref)


def test_classifier_eval_mode_and_fallbacks(monkeypatch):
    """eval: no mask, same kernels; more than 32 classes: the reference's convolution, with nn.Dropout2d doing the multiplication."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import ClassifierConv1x1, FoldedDropout2d
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 48, 4, 16, generator=g).to(dev)
    for Kc in (19, 40):
        conv = ClassifierConv1x1(48, Kc, kernel_size=1, bias=True).to(dev)
        net = nn.Sequential(FoldedDropout2d(0.5, conv), conv).to(dev)
        ref = nn.Sequential(nn.Dropout2d(0.5), nn.Conv2d(48, Kc, kernel_size=1, bias=True)).to(dev)
        ref[1].load_state_dict(conv.state_dict())
        net.eval(); ref.eval()
        with torch.no_grad():
            assert float((net(x) - ref(x)).abs().max()) <= 1e-5
        net.train(); ref.train()
        torch.manual_seed(3)
        a = net(x)
        torch.manual_seed(3)
        b = ref(x)
        assert float((a - b).abs().max()) <= 1e-5, Kc
        assert K.cls1x1_eligible(x, conv.weight) == (Kc <= 32)


@pytest.mark.parametrize("bn_training", [True, False])
def test_bias_gradient_of_a_convolution_in_front_of_batchnorm(bn_training, monkeypatch):
    """kernels.bias_grad: in front of a BatchNorm that uses the batch statistics the bias gradient is identically zero (the reference
    sums rounding noise); the BatchNorm backward marks its dx and the convolution returns zeros without a pass over dy. With frozen
    statistics (eval-mode BatchNorm) the sum is computed. Both against the reference's modules in fp64."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import HeadConv3x3, ModuleHelper
    dev = _dev()
    C = 96                                      # (a channel count the split weight-gradient kernel takes: kernels.CONV3X3_SB_WRW_CHANNELS)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(2, C, 8, 64, generator=g).relu_()
    dy0 = torch.randn(2, C, 8, 64, generator=g)
    torch.manual_seed(1)
    net = torch.nn.Sequential(HeadConv3x3(C), ModuleHelper.BNReLU(C, bn_type="torchsyncbn")).to(dev)
    ref = torch.nn.Sequential(nn.Conv2d(C, C, 3, 1, 1), nn.BatchNorm2d(C), nn.ReLU()).double().to(dev)
    ref[0].load_state_dict(net[0].state_dict())
    with torch.no_grad():
        rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
        for bn in [m for m in net.modules() if hasattr(m, "running_mean")] + [ref[1]]:
            bn.running_mean.copy_(rm.to(bn.running_mean.dtype))
            bn.running_var.copy_(rv.to(bn.running_var.dtype))
    net.train(); ref.train()
    if not bn_training:
        for m in list(net.modules()) + [ref[1]]:
            if hasattr(m, "running_mean"):
                m.eval()
    sums = []
    monkeypatch.setattr(K, "bias_grad", lambda dy, f=K.bias_grad: (sums.append(getattr(dy, "_cseg_zero_chan_sum", None) == dy._version), f(dy))[1])
    x = x0.to(dev).clone().requires_grad_(True)
    net(x).backward(dy0.to(dev))
    xr = x0.double().to(dev).clone().requires_grad_(True)
    ref(xr).backward(dy0.double().to(dev))
    got, want = net[0].bias.grad.double().cpu(), ref[0].bias.grad.cpu()
    scale = float(ref[0].weight.grad.abs().max())
    if K._on_device(x) and K.conv3x3_sb_eligible(x.detach(), net[0].weight) and K.conv3x3_sb_wrw_wanted(x.detach(), dy0.to(dev)):
        assert sums == [bn_training], "the BatchNorm backward must mark dx exactly when it used the batch statistics"
    if bn_training:
        assert float(want.abs().max()) <= 1e-9 * max(scale, 1.0)              # the identity the shortcut rests on, in fp64
        assert float(got.abs().max()) <= 1e-5 * scale
    else:
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_bottleneck_skip_gradient_in_the_epilogue_is_bit_identical(monkeypatch):
    """hrnet_backbone.Bottleneck without a downsample path: conv1 + the skip connection as one autograd node whose backward adds the skip
    gradient in the epilogue of the backward-data kernel (kernels.Conv1x1SplitSkip, cseg_conv1x1_split_fwd_add) against the two-node
    form with autograd's add (CSEG_SKIP_ADD_FUSED=0): the same two fp32 operands are added either way -- every gradient bit for bit."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import Bottleneck
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(2, 256, 6, 40, generator=g).relu_()
    dy0 = torch.randn(2, 256, 6, 40, generator=g)
    torch.manual_seed(4)
    blk = Bottleneck(256, 64, bn_type="torchsyncbn").to(dev).train()
    monkeypatch.setattr(K, "CONV1X1_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(K, "SKIP_ADD_FUSED", fused)
        calls = []
        orig = K.Conv1x1SplitSkip.apply
        monkeypatch.setattr(K.Conv1x1SplitSkip, "apply", staticmethod(lambda *a, o=orig: (calls.append(1), o(*a))[1]))
        state = {k: v.clone() for k, v in blk.state_dict().items()}
        x = x0.to(dev).clone().requires_grad_(True)
        blk.zero_grad()
        y = blk(x)
        y.backward(dy0.to(dev))
        blk.load_state_dict(state)                   # (running statistics back: the second pass starts from the same buffers)
        monkeypatch.setattr(K.Conv1x1SplitSkip, "apply", orig)
        assert bool(calls) == fused
        res.append([y.detach().cpu(), x.grad.cpu()] + [p.grad.cpu().clone() for p in blk.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # and the block against the reference's arithmetic in fp64
    ref = torch.nn.Sequential()
    xr = x0.double().to(dev).clone().requires_grad_(True)
    import torch.nn.functional as F
    w = [p.detach().double() for p in (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)]
    bns = [m for m in (blk.bn1, blk.bn2, blk.bn3)]

    def bn(t, m):
        mod = [q for q in m.modules() if hasattr(q, "running_mean")][0]
        mu, var = t.mean((0, 2, 3), keepdim=True), t.var((0, 2, 3), unbiased=False, keepdim=True)
        return (t - mu) / torch.sqrt(var + mod.eps) * mod.weight.detach().double().view(1, -1, 1, 1) + mod.bias.detach().double().view(1, -1, 1, 1)
    o = bn(F.conv2d(xr, w[0]), bns[0]).relu()
    o = bn(F.conv2d(o, w[1], padding=1), bns[1]).relu()
    o = (bn(F.conv2d(o, w[2]), bns[2]) + xr).relu()
    o.backward(dy0.double().to(dev))
    err = float((res[0][1].double() - xr.grad.cpu()).abs().max())
    assert err <= 2e-4 * float(xr.grad.abs().max()), err
