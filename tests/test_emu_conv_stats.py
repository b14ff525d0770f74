"""BatchNorm statistics from the convolution epilogue (csrc/cseg_stats.h) on the CPU emulation of the
execution model: for every split-operand forward kernel family, the (count, mean, M2) segment records the `_st` entry points write
must finalise (cseg_bn_tiles_finalize / cseg_bn_tiles_moments) to the same mean / invstd / running statistics / fp64 moments as the
statistics pass over the stored output (cseg_bn_stats_finalize / cseg_bn_stats) and as torch in fp64 -- ragged shapes included
(rows % 4 != 0, a last 64-pixel segment that is partly outside the tensor, H*W % 64 != 0) -- and a residual block built on them
must equal the block with the switch off, forward and backward."""
import pytest
import torch

from tests.emu import inject

CASES = [
    # kind, B, Cin, Cout, H, W, nt, bias
    ("c3", 2, 48, 48, 6, 96, 0, False),        # conv3x3_sb16p resident, second column tile half empty, rows % 4 = 2
    ("c3", 1, 96, 96, 5, 64, 0, False),        # conv3x3_sb_kernel<6>
    ("c3", 1, 64, 64, 4, 68, 0, False),        # conv3x3_sb16 (four channel tiles), 4 valid columns in the last segment
    ("c3", 1, 192, 192, 3, 32, 3, False),      # streamed weights, explicit tiling, half-empty tiles
    ("c3", 1, 144, 144, 9, 64, 0x109, True),   # the 8-row head kernel, with bias
    ("c1", 2, 64, 256, 5, 20, 0, False),       # 1x1: 100 flat pixels per image (one full segment + 36 pixels)
    ("c1", 1, 48, 144, 8, 64, 0, True),
    ("s2", 2, 48, 96, 5, 40, 0, False),        # stride 2: output 5 x 40
]


# the default route of every case + the 8-row tiles forced on the two cases the persistent small-channel kernels take
OPT_IN = [({}, c) for c in CASES] + [({"CSEG_SB16_ROWS8": "2"}, CASES[0]), ({"CSEG_SB16_ROWS8": "2"}, CASES[3])]


@pytest.mark.parametrize("env,case", OPT_IN)
def test_epilogue_statistics_equal_a_pass_over_the_output(env, case, monkeypatch):
    kind, B, Cin, Cout, H, W, nt, bias = case
    from contrastiveseg_amd import kernels as K
    inject.install(monkeypatch)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    g = torch.Generator().manual_seed(11 + Cin + H)
    k = 1 if kind == "c1" else 3
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    bvec = (torch.randn(Cout, generator=g) * 0.5) if bias else None
    if kind == "s2":
        x = torch.randn(B, Cin, 2 * H, 2 * W, generator=g) + 0.3
        y = K.conv3x3_s2_run(x, w, want_stats=True)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    elif kind == "c1":
        x = torch.randn(B, Cin, H, W, generator=g) + 0.3
        y = K.conv1x1_sb_run(x, w, False, bvec, want_stats=True)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None if bvec is None else bvec.double())
    else:
        x = torch.randn(B, Cin, H, W, generator=g) + 0.3
        y = K.conv3x3_sb_run(x, w, False, bvec, nt, want_stats=True)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None if bvec is None else bvec.double(), 1, 1)
    assert float((y.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    st = K.known_tile_stats(y)
    assert st is not None and st.shape[0] == Cout and st.shape[2] == 4
    n = y.numel() // Cout
    assert abs(float(st[:, :, 0].sum()) - Cout * n) < 0.5, "segment counts do not add up to the tensor"
    # finalised statistics vs the statistics pass over y and vs torch fp64 on the stored values
    rm0, rv0 = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    rm_a, rv_a, nb_a = rm0.clone(), rv0.clone(), torch.tensor(3)
    rm_b, rv_b, nb_b = rm0.clone(), rv0.clone(), torch.tensor(3)
    mi_t = K.bn_tiles_finalize(st, 1e-5, 0.1, rm_a, rv_a, nb_a)
    mi_p = K.bn_stats_finalize(y, 1e-5, 0.1, rm_b, rv_b, nb_b)
    yd = y.double().transpose(0, 1).reshape(Cout, -1)
    mean64, var64 = yd.mean(1), yd.var(1, unbiased=False)
    inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
    for mi in (mi_t, mi_p):
        assert float((mi[:, 0].double() - mean64).abs().max()) <= 2e-6 * max(1.0, float(mean64.abs().max()))
        assert float((mi[:, 1].double() / inv64 - 1).abs().max()) <= 2e-6
    assert int(nb_a) == int(nb_b) == 4
    assert float((rm_a - rm_b).abs().max()) <= 1e-6 and float((rv_a - rv_b).abs().max()) <= 1e-6 * float(rv_b.abs().max())
    mo_t, mo_p = K.bn_tiles_moments(st), K.bn_stats(y)
    assert mo_t.shape == mo_p.shape == (Cout + 1, 2) and float(mo_t[-1, 0]) == float(mo_p[-1, 0]) == n
    scale = float(mo_p[:-1].abs().max())
    assert float((mo_t[:-1] - mo_p[:-1]).abs().max()) <= 2e-6 * scale


def test_residual_block_with_epilogue_statistics_equals_the_plain_block(monkeypatch):
    """BasicBlock (conv -> bn -> relu -> conv -> bn -> + x -> relu, hrnet_backbone.py:49-65 of the reference) with the statistics taken
    from the convolution epilogues against the same block with CSEG_CONV_STATS off: outputs, running statistics and gradients."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    torch.manual_seed(5)
    blk = mark_conv_bn_pairs(BasicBlock(48, 48, bn_type="torchbn").train())
    assert blk.conv1.bn_follows and blk.conv2.bn_follows
    x0 = torch.randn(2, 48, 6, 64) + 0.2
    gy = torch.randn(2, 48, 6, 64)
    res = {}
    for on in (False, True):
        monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", on)
        for bn in (blk.bn1, blk.bn2):
            bn.reset_running_stats()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        y.backward(gy)
        monkeypatch.setattr(K._hip, "call", orig)
        res[on] = (y.detach(), x.grad.clone(), blk.conv1.weight.grad.clone(), blk.bn2.running_var.clone(), calls)
        blk.zero_grad()
    assert "cseg_bn_fwd_amax" in res[False][4] and "cseg_bn_tiles_finalize" not in res[False][4]
    assert res[True][4].count("cseg_bn_tiles_finalize") == 2 and res[True][4].count("cseg_conv3x3_split_fwd_st") == 2
    assert "cseg_bn_fwd_amax" not in res[True][4], "a statistics pass ran although the epilogue had the statistics"
    for a, b in zip(res[False][:4], res[True][:4]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("channels,hw", [(48, (6, 64)), (96, (4, 64)), (192, (4, 32))])
def test_fused_block_node_equals_the_four_nodes_it_replaces(channels, hw, monkeypatch):
    """kernels.BasicBlockSplit (one autograd node per residual block) against the unfused chain Conv3x3SplitFork -> _BNAct ->
    Conv3x3SplitBF16 -> _BNAct: the same kernel calls in the same order, so outputs, every gradient and the BN buffers are
    bit-identical; what changes is the number of autograd nodes."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    torch.manual_seed(channels)
    blk = mark_conv_bn_pairs(BasicBlock(channels, channels, bn_type="torchbn").train())
    x0 = torch.randn(2, channels, *hw) * 0.7 + 0.1
    gy = torch.randn(2, channels, *hw)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(K, "BLOCK_FUSED", fused)
        for bn in (blk.bn1, blk.bn2):
            bn.reset_running_stats()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        x = (x0.clone() * 1.0).requires_grad_(True)
        y = blk(x * 1.0)                        # a non-leaf input, as inside the network
        n_nodes = 0
        seen, stack = set(), [y.grad_fn]
        while stack:
            f = stack.pop()
            if f is None or f in seen:
                continue
            seen.add(f)
            n_nodes += 1
            stack += [g for g, _ in f.next_functions]
        y.backward(gy)
        monkeypatch.setattr(K._hip, "call", orig)
        res[fused] = ([y.detach(), x.grad.clone(), blk.conv1.weight.grad.clone(), blk.conv2.weight.grad.clone(), blk.bn1.weight.grad.clone(),
                       blk.bn2.bias.grad.clone(), blk.bn1.running_var.clone(), blk.bn2.running_mean.clone()], calls, n_nodes,
                      K.known_amax(y) is not None)
        blk.zero_grad()
    packs = ("cseg_amax_batch", "cseg_split_pack_batch")          # (the first use of a weight registers and packs it)
    assert [c for c in res[True][1] if c not in packs] == [c for c in res[False][1] if c not in packs], \
        "the fused node issues a different kernel sequence"
    assert res[True][2] < res[False][2], (res[True][2], res[False][2])
    assert res[True][3] and res[False][3], "the block output must carry its max|.| record for the next convolution"
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("channels", [[48, 96], [48, 96, 192]])
def test_fan_out_node_sums_the_branch_gradients_in_one_kernel(channels, monkeypatch):
    """kernels.fan_out (opt-in, CSEG_FANOUT_SUM=1): an exchange unit of HRNet with every branch output handed to its consumers through
    ONE autograd node whose backward sums the arriving gradients with the n-ary sum kernel -- same outputs bit for bit, input and
    parameter gradients equal to autograd's own accumulation up to the order of the additions, and the max|.| record of a branch
    output still reaches its consumers (no extra pass over the tensor)."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import HighResolutionModule
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    torch.manual_seed(len(channels))
    mod = mark_conv_bn_pairs(HighResolutionModule(channels, 1, "torchbn", 0.1).train())
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    xs0 = [torch.randn(2, c, 8 >> i, 64 >> i) * 0.5 for i, c in enumerate(channels)]
    gys = [torch.randn(2, c, 8 >> i, 64 >> i) for i, c in enumerate(channels)]
    res = {}
    for fan in (False, True):
        monkeypatch.setattr(K, "FANOUT_SUM", fan)
        mod.load_state_dict(state0)
        K.SPLIT_WEIGHTS.invalidate()
        mod.zero_grad()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        xs = [t.clone().requires_grad_(True) for t in xs0]
        ys = mod([t * 1.0 for t in xs])
        torch.autograd.backward(ys, gys)
        monkeypatch.setattr(K._hip, "call", orig)
        res[fan] = ([y.detach().clone() for y in ys], [x.grad.clone() for x in xs] + [p.grad.clone() for p in mod.parameters()], calls)
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b), "the forward must not change"
    for a, b in zip(res[False][1], res[True][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1e-3, float(a.abs().max()))
    n = len(channels)
    # one n-ary sum per branch (forward fuse sums: n launches either way); no additional max|.| passes
    assert res[True][2].count("cseg_fuse_sum_fwd") == res[False][2].count("cseg_fuse_sum_fwd") + n
    assert res[True][2].count("cseg_amax_f32") == res[False][2].count("cseg_amax_f32")


def test_fan_out_node_in_the_lockstep_exchange(monkeypatch):
    """The SyncBN form of the exchange unit (HighResolutionModule._exchange_lockstep: all conv+BN paths advanced side by side) with
    kernels.fan_out: same outputs and gradients as without it (single process: the grouped BN call degenerates to per-site calls)."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import HighResolutionModule
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    torch.manual_seed(5)
    channels = [48, 96, 192]
    mod = mark_conv_bn_pairs(HighResolutionModule(channels, 1, "torchbn", 0.1).train())
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    xs0 = [torch.randn(2, c, 8 >> i, 64 >> i) * 0.5 for i, c in enumerate(channels)]
    gys = [torch.randn(2, c, 8 >> i, 64 >> i) for i, c in enumerate(channels)]
    res = {}
    for fan in (False, True):
        monkeypatch.setattr(K, "FANOUT_SUM", fan)
        mod.load_state_dict(state0)
        K.SPLIT_WEIGHTS.invalidate()
        mod.zero_grad()
        xs = [t.clone().requires_grad_(True) for t in xs0]
        ys = mod._exchange_lockstep([t * 1.0 for t in xs])
        torch.autograd.backward(ys, gys)
        res[fan] = ([y.detach().clone() for y in ys], [x.grad.clone() for x in xs] + [p.grad.clone() for p in mod.fuse_layers.parameters()])
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[False][1], res[True][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1e-3, float(a.abs().max()))


# ---- round 5: widths / planes that are NOT multiples of 4 floats (DeepLab-R101-d8: 65 x 129; HRNet at 520 x 520: 130, 65, 33, 17) ----
RAGGED = [
    # kind, B, Cin, Cout, H, W, nt, bias, env
    ("c3", 1, 48, 48, 9, 65, 0, False, {}),                          # 8-row / 4-row persistent kernels, odd width, two column tiles
    ("c3", 1, 48, 48, 9, 65, 0, False, {"CSEG_SB16_ROWS8": "2"}),
    ("c3", 2, 96, 96, 5, 33, 0, True, {}),                           # conv3x3_sb_kernel<6>, with bias
    ("c3", 1, 64, 64, 6, 130, 0, False, {}),                         # W % 4 == 2: one-tile 16-channel-chunk kernel
    ("c3", 1, 192, 192, 3, 17, 3, False, {}),                        # streamed weights
    ("c3", 1, 144, 144, 9, 129, 0x109, True, {}),                    # the 8-row head kernel
    ("c1", 2, 64, 256, 5, 13, 0, True, {}),                          # plane 65
    ("c1", 1, 48, 144, 13, 43, 0, False, {}),                        # plane 559 = 2 x 256 + 47
]


@pytest.mark.parametrize("case", RAGGED, ids=lambda c: "%s-%d-%dx%d" % (c[0], c[2], c[4], c[5]))
def test_widths_that_are_not_multiples_of_four(case, monkeypatch):
    """The split-operand forward / backward-data kernels at odd widths: output, epilogue addend (3x3) and the statistics epilogue
    against torch in float64 -- the stores (and the addend loads) go element by element when a row is not 16-byte aligned
    (cseg_store_row4); the loaders never needed alignment."""
    kind, B, Cin, Cout, H, W, nt, bias, env = case
    from contrastiveseg_amd import kernels as K
    inject.install(monkeypatch)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    g = torch.Generator().manual_seed(3 + Cin + W)
    k = 1 if kind == "c1" else 3
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    bvec = (torch.randn(Cout, generator=g) * 0.5) if bias else None
    x = torch.randn(B, Cin, H, W, generator=g) + 0.3
    assert (H * W) % 4 != 0 if kind == "c1" else W % 4 != 0
    if kind == "c1":
        assert K.conv1x1_sb_eligible(x, w)
        y = K.conv1x1_sb_run(x, w, False, bvec, want_stats=True)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None if bvec is None else bvec.double())
        gy = torch.randn(B, Cout, H, W, generator=g)
        dx = K.conv1x1_sb_run(gy, w, True)
        dref = torch.nn.functional.conv_transpose2d(gy.double(), w.double())
    else:
        assert K.conv3x3_sb_eligible(x, w)
        y = K.conv3x3_sb_run(x, w, False, bvec, nt, want_stats=True)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None if bvec is None else bvec.double(), 1, 1)
        gy = torch.randn(B, Cout, H, W, generator=g)
        if nt == 0x109:
            dx = K.conv3x3_sb_run(gy, w, True, None, nt)
            dref = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), None, 1, 1)
        else:
            ad = torch.randn(B, Cin, H, W, generator=g)
            dx = K.conv3x3_sb_run(gy, w, True, None, nt, addend=ad)
            dref = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), None, 1, 1) + ad.double()
    assert not torch.isnan(y).any() and not torch.isnan(dx).any()
    assert float((y.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    assert float((dx.double() - dref).abs().max()) <= 3e-5 * float(dref.abs().max())
    st = K.known_tile_stats(y)
    n = y.numel() // Cout
    assert abs(float(st[:, :, 0].sum()) - Cout * n) < 0.5
    mi = K.bn_tiles_finalize(st, 1e-5, 0.1, None, None, None)
    yd = y.double().transpose(0, 1).reshape(Cout, -1)
    assert float((mi[:, 0].double() - yd.mean(1)).abs().max()) <= 2e-6 * max(1.0, float(yd.mean(1).abs().max()))
    assert float((mi[:, 1].double() * torch.sqrt(yd.var(1, unbiased=False) + 1e-5) - 1).abs().max()) <= 2e-6


def test_residual_block_at_an_odd_width_takes_the_split_kernels_for_forward_and_backward_data(monkeypatch):
    """A BasicBlock on 6 x 65 maps (the 1/8-resolution maps of HRNet at 520 x 520): forward, backward-data (incl. the identity gradient
    in its epilogue) and weight gradients of both convolutions on the split kernels -- as ONE autograd node, like at the aligned widths --
    against the same block on torch's own convolutions."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    torch.manual_seed(5)
    blk = mark_conv_bn_pairs(BasicBlock(48, 48, bn_type="torchbn").train())
    x0 = torch.randn(2, 48, 6, 65) + 0.2
    gy = torch.randn(2, 48, 6, 65)
    res = {}
    for split in (True, False):
        monkeypatch.setattr(K, "CONV3X3_SPLIT_BF16", split)
        for bn in (blk.bn1, blk.bn2):
            bn.reset_running_stats()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        y.backward(gy)
        monkeypatch.setattr(K._hip, "call", orig)
        res[split] = (y.detach(), x.grad.clone(), blk.conv1.weight.grad.clone(), blk.conv2.weight.grad.clone(), calls)
        blk.zero_grad()
    assert res[True][4].count("cseg_conv3x3_split_fwd_st") == 2 and "cseg_conv3x3_split_fwd_add" in res[True][4]
    assert res[True][4].count("cseg_conv3x3_split_wrw") == 2
    assert not any(n.startswith("cseg_conv3x3_split") for n in res[False][4])
    for a, b, what in zip(res[True][:4], res[False][:4], ("out", "dx", "dw1", "dw2")):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), what


@pytest.mark.parametrize("case", [(2, 48, 64, 5, 13), (1, 144, 160, 13, 43), (2, 64, 256, 9, 17), (1, 272, 144, 1, 33)])
def test_pointwise_weight_gradient_on_ragged_planes(case, monkeypatch):
    """cseg_conv1x1_split_wrw on planes that are not multiples of 32 pixels (round 5, the RAGGED loader: element loads clamped into the
    plane, zeros behind it): 65, 559, 153 and 33 pixels per plane, two channel blocks in both directions in the last case --
    against float64, twice (deterministic)."""
    from contrastiveseg_amd import kernels as K
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    B, ci, co, H, W = case
    assert (H * W) % 32 != 0
    g = torch.Generator().manual_seed(8 + W)
    x = torch.randn(B, ci, H, W, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    ref = torch.einsum("bohw,bchw->oc", dy.double(), x.double()).reshape(co, ci, 1, 1)
    assert K.conv1x1_sb_wrw_eligible(x, dy)
    got = K.conv1x1_sb_wrw(x, dy)
    assert not torch.isnan(got).any()
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got, K.conv1x1_sb_wrw(x, dy))


@pytest.mark.parametrize("case", [(2, 48, 48, 5, 65), (1, 64, 96, 9, 33), (1, 16, 48, 18, 130), (2, 96, 48, 3, 17)])
def test_weight_gradient_on_ragged_widths(case, monkeypatch):
    """cseg_conv3x3_split_wrw at widths that are not multiples of the row segment (round 5, RAGGED loaders): 65 = 64 + 1, 33 = 32 + 1,
    130 = 2 x 64 + 2 (with a run boundary: 18 rows), 17 < 32 -- against float64, twice (deterministic)."""
    from contrastiveseg_amd import kernels as K
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    B, ci, co, H, W = case
    g = torch.Generator().manual_seed(18 + W)
    x = torch.randn(B, ci, H, W, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), dy.double(), padding=1)
    assert K.conv3x3_sb_wrw_eligible(x, dy)
    got = K.conv3x3_sb_wrw(x, dy)
    assert not torch.isnan(got).any()
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), float((got.double() - ref).abs().max()) / float(ref.abs().max())
    assert torch.equal(got, K.conv3x3_sb_wrw(x, dy))


@pytest.mark.parametrize("case", [(1, 64, 64, 7, 65, 2, False), (2, 128, 64, 9, 33, 4, True), (1, 64, 128, 5, 130, 2, True), (1, 256, 256, 6, 20, 4, False)])
def test_dilated_convolution_matches_fp64(case, monkeypatch):
    """cseg_conv3x3_split_dil_fwd (round 5): rate 2 / 4, padding = rate, at ragged widths -- forward (with bias and the statistics
    epilogue), backward-data with an epilogue addend, and the autograd wrapper against torch in float64."""
    from contrastiveseg_amd import kernels as K
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    B, ci, co, H, W, d, bias = case
    g = torch.Generator().manual_seed(31 + W + d)
    x = torch.randn(B, ci, H, W, generator=g) + 0.2
    w = torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5
    b = torch.randn(co, generator=g) * 0.5 if bias else None
    assert K.conv3x3_dil_eligible(x, w, (d, d))
    y = K.conv3x3_dil_run(x, w, d, False, b, want_stats=True)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, d, d)
    assert not torch.isnan(y).any()
    assert float((y.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    st = K.known_tile_stats(y)
    mi = K.bn_tiles_finalize(st, 1e-5, 0.1, None, None, None)
    yd = y.double().transpose(0, 1).reshape(co, -1)
    assert float((mi[:, 0].double() - yd.mean(1)).abs().max()) <= 2e-6 * max(1.0, float(yd.mean(1).abs().max()))
    gy = torch.randn(B, co, H, W, generator=g)
    ad = torch.randn(B, ci, H, W, generator=g)
    dx = K.conv3x3_dil_run(gy, w, d, True, None, addend=ad)
    dref = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), None, 1, d, 0, 1, d) + ad.double()
    assert float((dx.double() - dref).abs().max()) <= 3e-5 * float(dref.abs().max())
    # autograd wrapper: dx on the split kernel, dw / db on the library convolution
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True) if bias else None
    K.conv3x3_dil_split(xg, wg, bg, d).backward(gy)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    torch.nn.functional.conv2d(x64, w64, b64, 1, d, d).backward(gy.double())
    assert float((xg.grad.double() - x64.grad).abs().max()) <= 3e-5 * float(x64.grad.abs().max())
    assert float((wg.grad.double() - w64.grad).abs().max()) <= 1e-4 * float(w64.grad.abs().max())
    if bias:
        assert float((bg.grad.double() - b64.grad).abs().max()) <= 1e-4 * float(b64.grad.abs().max())
