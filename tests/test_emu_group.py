"""Grouped launches (csrc/conv3x3_group.hip, the cseg_*_group_* entry points of include/cseg_hip.h) on the CPU emulation of the
execution model: ONE launch over several independent layers must give, member by member, the BITS of the one-layer entry points
(same tile body, same K order per output element -- whichever block computes a tile), and agree with float64 convolutions.
Reference shape of the work: the parallel branches of an HRNet exchange unit, lib/models/backbones/hrnet/hrnet_backbone.py:262-288."""
import os

import numpy as np
import pytest

from tests.emu import build_emu
from tests.emu import harness as E

pytestmark = pytest.mark.skipif(not os.path.exists(build_emu.CLANG), reason="host clang++ of the ROCm toolchain not found")


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _bound(ref, k_len):
    return 3e-6 * np.sqrt(k_len) * max(1.0, float(np.abs(ref).max()))


@pytest.fixture(params=["asc", "desc"])
def wave_order(request, monkeypatch):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", request.param)
    return request.param


@pytest.fixture(params=["tile4", "tile8", "tile8pc", "tile8pc12"])
def form(request, monkeypatch):
    """The tile forms of the grouped convolution kernel (the library picks by the size of the group; forced here): 256-pixel tiles,
    512-pixel tiles with all waves in the same phases, with 4 producer + 4 consumer waves, with 8 computing + 4 staging waves."""
    monkeypatch.setenv("CSEG_GROUP_TILE", "4" if request.param == "tile4" else "8")
    monkeypatch.setenv("CSEG_GROUP_PC", {"tile8pc": "1", "tile8pc12": "2"}.get(request.param, "0"))
    return request.param


# (B, Cin, Cout, H, W) per member: the channel ladder of HRNet's branches on small maps -- resident (48) and streamed (>= 64) operators,
# several channel groups per tile, ragged tiles, a width that is not a multiple of 4, fewer tiles than XCDs
GROUPS = [
    [(1, 48, 48, 9, 70), (1, 96, 96, 5, 36)],
    [(2, 48, 48, 4, 64), (2, 96, 96, 6, 33), (1, 192, 48, 3, 20), (1, 64, 144, 2, 8)],
    [(1, 32, 48, 5, 17)],
]


@pytest.mark.parametrize("shapes", GROUPS)
def test_group_forward_equals_the_one_layer_launches_bit_for_bit(shapes, wave_order, form):
    members = []
    for i, (B, ci, co, H, W) in enumerate(shapes):
        members.append(dict(x=_rand((B, ci, H, W), 10 + i, 1.0 + i), w=_rand((co, ci, 3, 3), 20 + i, 1.0 / (3 * ci ** 0.5)),
                            bias=_rand((co,), 30 + i) if i % 2 else None, stats=True))
    outs = E.conv3x3_group(members)
    for m, (y, st) in zip(members, outs):
        y1, st1 = E.conv3x3_sb_st(m["x"], m["w"], m["bias"], nt=E.NT_GROUP)
        assert not np.isnan(y).any() and not np.isnan(st[..., :3]).any()
        assert np.array_equal(y, y1)
        # (count, mean, M2) per row segment: this kernel's own fixed summation order -- equal to the one-layer kernel's to rounding
        assert np.array_equal(st[..., 0], st1[..., 0])
        assert np.abs(st[..., 1] - st1[..., 1]).max() <= 2e-6 * max(1.0, np.abs(st1[..., 1]).max())
        assert np.abs(st[..., 2] - st1[..., 2]).max() <= 1e-5 * max(1.0, np.abs(st1[..., 2]).max())
        ref = E.ref_conv3x3(m["x"], m["w"], m["bias"])
        assert np.abs(y - ref).max() <= _bound(ref, 9 * m["x"].shape[1])


def test_group_backward_data_with_addend_and_a_reused_scheduling_record(wave_order, form):
    """The backward-data operators of a depth (transposed packing), one of them with the residual gradient added in the epilogue; the
    same scheduling record serves two launches (each leaves it zero)."""
    shapes = [(1, 48, 48, 6, 40), (1, 96, 96, 3, 24)]
    members = []
    for i, (B, c, _, H, W) in enumerate(shapes):
        members.append(dict(x=_rand((B, c, H, W), 40 + i), w=_rand((c, c, 3, 3), 50 + i, 1.0 / (3 * c ** 0.5)), transpose_flip=True,
                            addend=_rand((B, c, H, W), 60 + i) if i == 0 else None))
    sched = E.aligned_to((320,), np.int32, 128)
    for _ in range(2):
        outs = E.conv3x3_group(members, sched=sched)
        for m, (y, _st) in zip(members, outs):
            y1 = E.conv3x3_sb(m["x"], m["w"], None, transpose_flip=True, nt=E.NT_GROUP, arith=E.F16X3, addend=m["addend"])
            assert np.array_equal(y, y1)
            ref = E.ref_conv3x3_bwd_data(m["x"], m["w"]) + (0 if m["addend"] is None else m["addend"])
            assert np.abs(y - ref).max() <= _bound(ref, 9 * m["x"].shape[1])


def test_group_rejects_what_it_does_not_cover():
    x, w = _rand((1, 16, 4, 8), 1), _rand((48, 16, 3, 3), 2)
    with pytest.raises(RuntimeError, match="Cin >= 32"):
        E.conv3x3_group([dict(x=x, w=w)])


# ---- the autograd node on top of the grouped launches -------------------------------------------------------------------------
import torch

from tests.emu import inject


@pytest.mark.parametrize("channels,shapes", [((48, 96), ((6, 40), (3, 20))), ((48, 96, 192), ((8, 64), (4, 32), (2, 16)))])
def test_grouped_depth_node_equals_the_per_branch_nodes_bit_for_bit(channels, shapes, form, monkeypatch):
    """kernels.BasicBlockGroup (ONE autograd node for the residual blocks of a depth of HRNet's parallel branches, on the grouped
    conv / BatchNorm / weight-gradient launches) against one kernels.BasicBlockSplit per branch with the group's tile body
    (nt = CSEG_NT_GROUP): outputs, every gradient, running statistics, num_batches_tracked and the max|.| records are bit-identical
    over two consecutive steps; the grouped node issues 12 library calls (16 launches) per depth whatever the number of branches."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    monkeypatch.setattr(K, "_GROUP_SCHED", {})
    # the per-branch nodes on the group's tile body: every channel count asks for nt = CSEG_NT_GROUP
    monkeypatch.setattr(K, "CONV3X3_SB_PICK_NT_CHANNELS", tuple(channels))
    monkeypatch.setattr(K, "conv3x3_sb_pick_nt", lambda x, c: K.NT_GROUP)
    monkeypatch.setattr(K, "conv3x3_sb_tiles", lambda x, c: 1 << 20)          # (the routing threshold counts blocks of the default tiling)
    torch.manual_seed(sum(channels))
    blocks = [mark_conv_bn_pairs(BasicBlock(c, c, bn_type="torchbn").train()) for c in channels]
    x0 = [torch.randn(2, c, *hw) * 0.7 + 0.1 for c, hw in zip(channels, shapes)]
    gy = [torch.randn(2, c, *hw) for c, hw in zip(channels, shapes)]
    res = {}
    for grouped in (False, True):
        for blk in blocks:
            blk.zero_grad()
            for bn in (blk.bn1, blk.bn2):
                bn.reset_running_stats()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        out = []
        for step in range(2):
            xs = [(x.clone() * 1.0).requires_grad_(True) for x in x0]
            ins = [x * 1.0 for x in xs]                  # non-leaf inputs, as inside the network
            if grouped:
                ys = K.basic_block_group(blocks, ins)
                assert ys is not None
            else:
                ys = [K.basic_block_split(xi, blk) for xi, blk in zip(ins, blocks)]
            assert all(K.known_amax(y) is not None for y in ys)
            torch.autograd.backward(ys, gy)
            for y, x, blk in zip(ys, xs, blocks):
                out += [y.detach().clone(), x.grad.clone(), blk.conv1.weight.grad.clone(), blk.conv2.weight.grad.clone(),
                        blk.bn1.weight.grad.clone(), blk.bn1.bias.grad.clone(), blk.bn2.weight.grad.clone(), blk.bn2.bias.grad.clone(),
                        blk.bn1.running_mean.clone(), blk.bn1.running_var.clone(), blk.bn2.running_mean.clone(), blk.bn2.running_var.clone(),
                        blk.bn1.num_batches_tracked.clone(), blk.bn2.num_batches_tracked.clone(), K.known_amax(y).clone()]
        monkeypatch.setattr(K._hip, "call", orig)
        res[grouped] = (out, [c for c in calls if c not in ("cseg_amax_batch", "cseg_split_pack_batch", "cseg_amax_f32")])
    assert len(res[True][1]) == 2 * 12, res[True][1]                  # 12 library calls (16 launches) per depth and step
    assert len(res[False][1]) == 2 * 12 * len(channels), len(res[False][1])
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert torch.equal(a, b), (i % 15, i // 15)


def test_exchange_unit_with_grouped_batchnorm_equals_the_per_site_form(monkeypatch):
    """A three-branch HighResolutionModule in training mode, single rank, on the emulated device: branches on the grouped launches and
    the exchange unit depth by depth with its BatchNorm sites on the grouped launches (fused_bn._BNActGroupLocal), against the same
    module with the per-site BatchNorm calls (CSEG_EXCHANGE_GROUPED off): outputs, input and parameter gradients, running statistics
    bit-identical; the grouped form issues fewer library calls."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    inject.install(monkeypatch)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV1X1_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    monkeypatch.setattr(K, "_GROUP_SCHED", {})
    torch.manual_seed(7)
    mod = mark_conv_bn_pairs(HB.HighResolutionModule([48, 96, 192], 1, "torchbn", 0.1).train())
    g = torch.Generator().manual_seed(3)
    x0 = [torch.randn(2, 48, 16, 64, generator=g) + 0.1, torch.randn(2, 96, 8, 32, generator=g) + 0.1, torch.randn(2, 192, 4, 16, generator=g) + 0.1]
    res = {}
    for grouped in (False, True):
        monkeypatch.setattr(HB, "EXCHANGE_GROUPED", grouped)
        mod.zero_grad()
        for m in mod.modules():
            if hasattr(m, "reset_running_stats"):
                m.reset_running_stats()
        calls = []
        orig = K._hip.call
        monkeypatch.setattr(K._hip, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        xs = [x.clone().requires_grad_(True) for x in x0]
        outs = mod([x * 1.0 for x in xs])
        sum((o * o).mean() for o in outs).backward()
        monkeypatch.setattr(K._hip, "call", orig)
        res[grouped] = ([o.detach().clone() for o in outs] + [x.grad.clone() for x in xs]
                        + [p.grad.clone() for p in mod.parameters()] + [b.clone() for b in mod.buffers()], calls)
    assert any(c == "cseg_bn_group_apply" for c in res[True][1])
    n_bn = lambda cs: sum(1 for c in cs if c.startswith("cseg_bn"))
    assert n_bn(res[True][1]) < n_bn(res[False][1]), (n_bn(res[True][1]), n_bn(res[False][1]))
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
