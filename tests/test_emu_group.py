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


# (B, Cin, Cout, H, W) per member: the channel ladder of HRNet's branches on small maps -- resident (48) and streamed (>= 64) operators,
# several channel groups per tile, ragged tiles, a width that is not a multiple of 4, fewer tiles than XCDs
GROUPS = [
    [(1, 48, 48, 9, 70), (1, 96, 96, 5, 36)],
    [(2, 48, 48, 4, 64), (2, 96, 96, 6, 33), (1, 192, 48, 3, 20), (1, 64, 144, 2, 8)],
    [(1, 32, 48, 5, 17)],
]


@pytest.mark.parametrize("shapes", GROUPS)
def test_group_forward_equals_the_one_layer_launches_bit_for_bit(shapes, wave_order):
    members = []
    for i, (B, ci, co, H, W) in enumerate(shapes):
        members.append(dict(x=_rand((B, ci, H, W), 10 + i, 1.0 + i), w=_rand((co, ci, 3, 3), 20 + i, 1.0 / (3 * ci ** 0.5)),
                            bias=_rand((co,), 30 + i) if i % 2 else None, stats=True))
    outs = E.conv3x3_group(members)
    for m, (y, st) in zip(members, outs):
        y1, st1 = E.conv3x3_sb_st(m["x"], m["w"], m["bias"], nt=E.NT_GROUP)
        assert not np.isnan(y).any() and not np.isnan(st[..., :3]).any()
        assert np.array_equal(y, y1)
        assert np.array_equal(st[..., :3], st1[..., :3])
        ref = E.ref_conv3x3(m["x"], m["w"], m["bias"])
        assert np.abs(y - ref).max() <= _bound(ref, 9 * m["x"].shape[1])


def test_group_backward_data_with_addend_and_a_reused_scheduling_record(wave_order):
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
