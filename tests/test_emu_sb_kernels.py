"""The split-bf16 kernel SOURCES (contrastiveseg_amd/csrc/conv3x3_sb.hip, conv3x3_sb_wrw.hip, conv1x1_sb.hip,
conv1x1_sb_wrw.hip) executed on the CPU emulation of wave64 / LDS / MFMA in tests/emu, against float64 convolutions.

Two groups:
  * kernels that HAVE passed parity on an MI355X (3x3 forward / backward-data with both B-staging variants, weight gradient
    version 1): they pin the emulator -- MFMA operand layout, LDS-DMA addressing, barrier semantics;
  * kernels written after the round's GPU budget was spent (weight gradient version 2 with producer / consumer waves,
    the 1x1 forward / backward-data kernel, the 1x1 weight gradient, the explicit channel tilings): functionally verified
    here, source line for source line, before their first hardware run.
Every case also runs with the waves of a block scheduled in descending order: a result that depends on the order in which
waves reach a point between two barriers is a missing barrier."""
import os

import numpy as np
import pytest

from tests.emu import build_emu
from tests.emu import harness as E

pytestmark = pytest.mark.skipif(not os.path.exists(build_emu.CLANG), reason="host clang++ of the ROCm toolchain not found")


def _bound(ref, k_len):
    """fp32-class bound: split-bf16 keeps terms down to 2^-16 of the leading one; accumulation is fp32 over k_len products."""
    return 3e-6 * np.sqrt(k_len) * max(1.0, float(np.abs(ref).max()))


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@pytest.fixture(params=["asc", "desc"])
def wave_order(request, monkeypatch):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", request.param)
    return request.param


FWD_CASES = [  # B, Cin, Cout, H, W
    (1, 48, 48, 6, 68),        # ragged tiles both ways, 16-channel tail (48 = 32 + 16), NT = 3
    (2, 16, 96, 4, 64),        # tail-only K loop, NT = 6
    (1, 64, 144, 5, 24),       # two full chunks, NT = 9, narrow map
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("glds,var", [("1", "0"), ("0", "0"), ("1", "1"), ("1", "2")])
def test_conv3x3_forward_and_backward_data(case, glds, var, wave_order, monkeypatch):
    """var 0: the hardware-verified kernel (pins the emulator). var 1: the buffer-load addressing of the patch (no spills at
    9 channel tiles per block); var 2: conv3x3_sb16.hip (16-channel chunks, two blocks per CU) for up to 192 output channels.
    Both: first hardware run pending."""
    monkeypatch.setenv("CSEG_CONV3X3_SB_GLDS", glds)
    monkeypatch.setenv("CSEG_CONV3X3_SB_VAR", var)
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 1), _rand((co, ci, 3, 3), 2, 1.0 / (3 * ci ** 0.5)), _rand((co,), 3)
    y = E.conv3x3_sb(x, w, b)
    ref = E.ref_conv3x3(x, w, b)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)
    if ci % 48 == 0 and co % 16 == 0:
        dy = _rand((B, co, H, W), 4)
        dx = E.conv3x3_sb(dy, w, None, transpose_flip=True)
        ref = E.ref_conv3x3_bwd_data(dy, w)
        assert np.abs(dx - ref).max() <= _bound(ref, 9 * co)


@pytest.mark.parametrize("var", ["0", "2"])
@pytest.mark.parametrize("nt", [3, 6])
def test_conv3x3_explicit_channel_tiling(nt, var, wave_order, monkeypatch):
    monkeypatch.setenv("CSEG_CONV3X3_SB_VAR", var)
    """cseg_conv3x3_sb_*_nt (first hardware run pending): same convolution whatever the channel tiles per block."""
    B, ci, co, H, W = 1, 48, 96, 5, 40
    x, w = _rand((B, ci, H, W), 5), _rand((co, ci, 3, 3), 6, 1.0 / (3 * ci ** 0.5))
    y = E.conv3x3_sb(x, w, None, nt=nt)
    ref = E.ref_conv3x3(x, w)
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)
    assert np.array_equal(y, E.conv3x3_sb(x, w, None)) or np.abs(y - E.conv3x3_sb(x, w, None)).max() <= 1e-5


WRW_CASES = [  # B, Cin, Cout, H, W
    (1, 48, 48, 5, 64),        # ragged 64-wide channel block (3 of 4 tiles), one run
    (2, 16, 48, 9, 128),       # two column segments, several runs / units per split
    (1, 80, 96, 3, 64),        # two channel blocks each way
]


@pytest.mark.parametrize("case", WRW_CASES)
@pytest.mark.parametrize("version", ["1", "2"])
def test_conv3x3_weight_gradient(case, version, wave_order, monkeypatch):
    """Version 1 is hardware-verified; version 2 (producer / consumer waves, 4-slot ring) is verified here first."""
    monkeypatch.setenv("CSEG_CONV3X3_SB_WRW_V", version)
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 7), _rand((B, co, H, W), 8)
    dw = E.conv3x3_sb_wrw(x, dy)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= _bound(ref, B * H * W)


ONE_CASES = [  # B, Cin, Cout, H, W
    (1, 48, 64, 10, 30),       # ragged pixel tile (300 pixels), 16-channel tail, NT = 4
    (2, 144, 48, 16, 16),      # NT = 3, five K-steps with a tail
    (1, 64, 256, 4, 20),       # NT = 8 (the projection head's second layer), 80 pixels
    (1, 32, 144, 6, 44),       # NT = 9
    (1, 48, 512, 3, 44),       # f16x3: NT = 16 (round 5, wide tiling), ragged pixel tile; bf16x6: NT = 8, two channel groups
    (1, 32, 240, 4, 20),       # f16x3: NT = 15 (the 720-channel head's tiling); bf16x6: NT = 3, five channel groups
    (1, 512, 48, 2, 36),       # the backward-data operator of a 48 -> 512 layer: 16 tiles on the transposed weights
]


@pytest.mark.parametrize("case", ONE_CASES)
def test_conv1x1_forward_and_backward_data(case, wave_order):
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 9), _rand((co, ci, 1, 1), 10, 1.0 / ci ** 0.5), _rand((co,), 11)
    y = E.conv1x1_sb(x, w, b)
    ref = np.einsum("bchw,oc->bohw", x.astype(np.float64), w[:, :, 0, 0].astype(np.float64)) + b.astype(np.float64)[None, :, None, None]
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, ci)
    if ci % 48 and ci % 64:
        return                                   # backward-data needs the INPUT channel count to tile (48 | 64)
    dy = _rand((B, co, H, W), 12)
    dx = E.conv1x1_sb(dy, w, None, transpose=True)
    ref = np.einsum("bohw,oc->bchw", dy.astype(np.float64), w[:, :, 0, 0].astype(np.float64))
    assert not np.isnan(dx).any()
    assert np.abs(dx - ref).max() <= _bound(ref, co)


# the last two: 1 280 / 1 315 stages = five and more per split at the 256 splits the library then takes, so that the loaders' four-deep register ring of round 6 goes round (aligned
# planes; and a ragged one -- 65 x 129 -- whose last stage of an image is partly outside the plane while younger stages are in flight)
@pytest.mark.parametrize("case", [(2, 48, 64, 8, 8), (1, 144, 160, 8, 12), (2, 64, 256, 16, 16), (1, 16, 16, 4, 8),
                                  (1, 16, 16, 160, 256), (5, 16, 16, 65, 129)])
def test_conv1x1_weight_gradient(case, wave_order):
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 13), _rand((B, co, H, W), 14)
    dw = E.conv1x1_sb_wrw(x, dy)
    ref = np.einsum("bohw,bchw->oc", dy.astype(np.float64), x.astype(np.float64)).reshape(co, ci, 1, 1)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= _bound(ref, B * H * W)


# ---- the head's channel count (720 = 22 x 32 + 16: many chunks, then the tail), one spatial tile each ------------------
@pytest.mark.slow
def test_head_width_one_tile_each(monkeypatch):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", "shuffle:7")
    C = 720
    x, w, b = _rand((1, C, 4, 64), 21), _rand((C, C, 3, 3), 22, 1.0 / (3 * C ** 0.5)), _rand((C,), 23)
    ref = E.ref_conv3x3(x, w, b)
    for var in ("0", "1"):                            # 1 = buffer-load addressing of the patch
        monkeypatch.setenv("CSEG_CONV3X3_SB_VAR", var)
        y = E.conv3x3_sb(x, w, b)
        assert np.abs(y - ref).max() <= _bound(ref, 9 * C)
    monkeypatch.delenv("CSEG_CONV3X3_SB_VAR")
    w1, w2 = _rand((C, C, 1, 1), 24, 1.0 / C ** 0.5), _rand((256, C, 1, 1), 25, 1.0 / C ** 0.5)
    for wt in (w1, w2):                               # projection head: 720 -> 720 (NT = 9), 720 -> 256 (NT = 8)
        y = E.conv1x1_sb(x, wt, None)
        ref = np.einsum("bchw,oc->bohw", x.astype(np.float64), wt[:, :, 0, 0].astype(np.float64))
        assert np.abs(y - ref).max() <= _bound(ref, C)
    dz = _rand((1, 256, 4, 64), 26)
    dx = E.conv1x1_sb(dz, w2, None, transpose=True)    # 256 -> 720
    ref = np.einsum("bohw,oc->bchw", dz.astype(np.float64), w2[:, :, 0, 0].astype(np.float64))
    assert np.abs(dx - ref).max() <= _bound(ref, 256)
    dw = E.conv1x1_sb_wrw(x, dz)
    ref = np.einsum("bohw,bchw->oc", dz.astype(np.float64), x.astype(np.float64)).reshape(256, C, 1, 1)
    assert np.abs(dw - ref).max() <= _bound(ref, 256)


@pytest.mark.slow
@pytest.mark.parametrize("version", ["1", "2"])
def test_head_width_weight_gradient(version, monkeypatch):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", "shuffle:3")
    monkeypatch.setenv("CSEG_CONV3X3_SB_WRW_V", version)
    C = 720
    x, dy = _rand((1, C, 3, 64), 27), _rand((1, C, 3, 64), 28)
    dw = E.conv3x3_sb_wrw(x, dy)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= _bound(ref, 3 * 64)


# ---- round 3: the same kernels through the selectable-arithmetic entry points (cseg_*_split_*) -------------------------------
# f16x3 = two scaled fp16 pieces, three MFMAs per product. The operands get magnitudes far from 1 (activations ~1e3, gradients
# ~1e-7) so that a wrong or missing power-of-two scale shows: unscaled fp16 would overflow / flush them.
ARITHS = [E.BF16X6, E.F16X3]


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("case,var", [(FWD_CASES[0], "auto"), (FWD_CASES[1], "auto"), (FWD_CASES[2], "auto"), (FWD_CASES[0], "1")])
def test_split_conv3x3_forward_and_backward_data(case, var, arith, wave_order, monkeypatch):
    if var != "auto":
        monkeypatch.setenv("CSEG_CONV3X3_SB_VAR", var)
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 31, 1e3), _rand((co, ci, 3, 3), 32, 1e-4 / (3 * ci ** 0.5)), _rand((co,), 33, 0.1)
    y = E.conv3x3_sb(x, w, b, arith=arith)
    ref = E.ref_conv3x3(x, w, b)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)
    if ci % 48 == 0 and co % 16 == 0:
        dy = _rand((B, co, H, W), 34, 1e-7)
        dx = E.conv3x3_sb(dy, w, None, transpose_flip=True, arith=arith)
        ref = E.ref_conv3x3_bwd_data(dy, w)
        assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(9 * co) * float(np.abs(ref).max())


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("case", WRW_CASES)
def test_split_conv3x3_weight_gradient(case, arith, wave_order):
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 37, 50.0), _rand((B, co, H, W), 38, 1e-6)
    dw = E.conv3x3_sb_wrw(x, dy, arith=arith)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * H * W) * float(np.abs(ref).max())


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("case", [(2, 48, 96, 5, 32), (1, 80, 48, 9, 32), (1, 64, 48, 19, 96)])
def test_split_weight_gradient_32_pixel_segments(case, arith, wave_order):
    """widths that are 32 mod 64 (the 16 x 32 maps of the 384-channel branch): one K-step per row-step, runs of 8 rows"""
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 61, 7.0), _rand((B, co, H, W), 62, 1e-4)
    dw = E.conv3x3_sb_wrw(x, dy, arith=arith)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * H * W) * float(np.abs(ref).max())


@pytest.mark.parametrize("case", [(1, 64, 64, 5, 64), (2, 32, 128, 3, 32), (1, 48, 16, 4, 65), (1, 16, 256, 2, 64)])
def test_split_weight_gradient_partly_filled_channel_block(case, wave_order):
    """Round 6, f16x3: output channel counts that are multiples of 16 but not of 48 (the 64-channel 3x3 convolutions of HRNet's layer 1,
    ResNet's 64 / 128 / 256): the last 48-channel block holds 16 or 32 channels; its missing rows are neither staged nor stored."""
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 71, 3.0), _rand((B, co, H, W), 72, 1e-3)
    dw = E.conv3x3_sb_wrw(x, dy, arith=E.F16X3)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * H * W) * float(np.abs(ref).max())


@pytest.mark.parametrize("case", [(1, 192, 192, 4, 64), (1, 96, 192, 4, 64)])
def test_split_weight_gradient_group_order(case, wave_order):
    """several channel blocks each way: the XCD-aware block order (groups of SC x SI channel blocks) covers every block once."""
    B, ci, co, H, W = case
    x, dy = _rand((B, ci, H, W), 41), _rand((B, co, H, W), 42)
    ref = E.ref_conv3x3_wrw(x, dy)
    for arith in ARITHS:
        dw = E.conv3x3_sb_wrw(x, dy, arith=arith)
        assert not np.isnan(dw).any()
        assert np.abs(dw - ref).max() <= _bound(ref, B * H * W)


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("case", ONE_CASES)
def test_split_conv1x1(case, arith, wave_order):
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 43, 300.0), _rand((co, ci, 1, 1), 44, 1e-3 / ci ** 0.5), _rand((co,), 45, 0.1)
    y = E.conv1x1_sb(x, w, b, arith=arith)
    ref = np.einsum("bchw,oc->bohw", x.astype(np.float64), w[:, :, 0, 0].astype(np.float64)) + b.astype(np.float64)[None, :, None, None]
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, ci)
    if not (ci % 48 and ci % 64):
        dy = _rand((B, co, H, W), 46, 1e-8)
        dx = E.conv1x1_sb(dy, w, None, transpose=True, arith=arith)
        ref = np.einsum("bohw,oc->bchw", dy.astype(np.float64), w[:, :, 0, 0].astype(np.float64))
        assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(co) * float(np.abs(ref).max())
    if (H * W) % 32 == 0:
        dy = _rand((B, co, H, W), 47, 1e-5)
        dw = E.conv1x1_sb_wrw(x, dy, arith=arith)
        ref = np.einsum("bohw,bchw->oc", dy.astype(np.float64), x.astype(np.float64)).reshape(co, ci, 1, 1)
        assert np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * H * W) * float(np.abs(ref).max())


def test_f16x3_error_is_fp32_class():
    """Same operands through both arithmetics: the f16x3 error vs float64 stays within 2x of the bf16x6 error plus the fp32
    accumulation floor (the two share the accumulation order; what differs is the 2^-22 vs 2^-24 operand representation)."""
    B, ci, co, H, W = 1, 144, 48, 4, 64
    x, w = _rand((B, ci, H, W), 51), _rand((co, ci, 3, 3), 52, 1.0 / (3 * ci ** 0.5))
    ref = E.ref_conv3x3(x, w)
    e6 = np.abs(E.conv3x3_sb(x, w, None, arith=E.BF16X6) - ref).max()
    e3 = np.abs(E.conv3x3_sb(x, w, None, arith=E.F16X3) - ref).max()
    x32 = np.abs(E.ref_conv3x3(x, w).astype(np.float32) - ref).max()         # one fp32 rounding of the exact result
    assert e3 <= 2.0 * e6 + 4.0 * x32, (e3, e6, x32)


# ---- round 3: the persistent chunk-barrier form of the 16-channel-chunk kernel (conv3x3_sb16p_kernel, f16x3) ------------------
@pytest.mark.parametrize("case,what", [
    ((1, 96, 48, 5, 68), "weights streamed chunk by chunk (6 chunks do not fit), ragged tiles"),
    ((1, 64, 64, 6, 68), "4 channel tiles per block (layer-1 bottleneck), streamed"),
    ((1, 32, 48, 104, 640), "260 tiles on 256 blocks: some blocks walk two tiles with the weights resident"),
    ((2, 64, 64, 52, 640), "260 tiles, streamed weights across the tile boundary"),
])
def test_persistent_small_channel_convolution(case, what, monkeypatch):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", "shuffle:11")
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 71, 2.0), _rand((co, ci, 3, 3), 72, 1.0 / (3 * ci ** 0.5)), _rand((co,), 73)
    y = E.conv3x3_sb(x, w, b, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w, b)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci), what
    monkeypatch.setenv("CSEG_CONV3X3_SB16_P", "0")                 # the one-tile kernel on the same operands: same arithmetic
    y1 = E.conv3x3_sb(x, w, b, arith=E.F16X3)
    assert np.array_equal(y, y1), "persistent and one-tile kernels accumulate in the same order: bit-identical results"


# ---- 3x3 / stride 2 / pad 1 (csrc/conv3x3_s2.hip and the stride-2 weight gradient of conv3x3_sb_wrw.hip), f16x3 -----------------
S2_FWD_CASES = [  # B, Cin, Cout, Ho, Wo, nt
    (1, 48, 48, 5, 32, 3),       # ragged row tile, half a column tile
    (2, 32, 96, 3, 68, 6),       # two column tiles, the second ragged; six channel tiles per block
    (1, 16, 96, 9, 132, 3),      # one chunk, three row tiles, two channel tile groups
    (1, 64, 64, 6, 64, 4),       # round 6: 64 output channels = four tiles per block (the second stem convolution of HRNet)
    (2, 16, 128, 3, 36, 4),      # two 64-channel groups
]


@pytest.mark.parametrize("case", S2_FWD_CASES)
def test_stride2_forward(case, wave_order):
    B, ci, co, Ho, Wo, nt = case
    x, w = _rand((B, ci, 2 * Ho, 2 * Wo), 71, 3.0), _rand((co, ci, 3, 3), 72, 0.1)
    y = E.conv3x3_s2(x, w, nt)
    ref = E.ref_conv3x3_s2(x, w)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)


@pytest.mark.parametrize("case", [(1, 48, 48, 5, 32, 3), (2, 96, 32, 3, 66, 6), (1, 96, 16, 9, 130, 3)])
def test_stride2_backward_data(case, wave_order):
    """every pixel of dx is written exactly once (NaN-filled output buffer), all four parity classes, ragged quad tiles"""
    B, ci, co, Ho, Wo, nt = case
    dy, w = _rand((B, co, Ho, Wo), 73, 1e-3), _rand((co, ci, 3, 3), 74, 0.1)
    dx = E.conv3x3_s2_bwd(dy, w, nt)
    ref = E.ref_conv3x3_s2_bwd_data(dy, w)
    assert not np.isnan(dx).any()
    assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(4 * co) * float(np.abs(ref).max())


# the last two (round 6): output channels % 16, not % 48 -- a partly filled last 48-channel block (64 = 48 + 16, 128 = 2 x 48 + 32)
@pytest.mark.parametrize("case", [(1, 48, 48, 5, 32), (2, 80, 96, 3, 64), (1, 16, 48, 18, 32), (1, 192, 96, 2, 32),
                                  (1, 64, 64, 5, 32), (1, 16, 128, 3, 64)])
@pytest.mark.parametrize("rpu", ["4", "16"])
def test_stride2_weight_gradient(case, rpu, wave_order, monkeypatch):
    """runs of 4 / 16 rows (the benched shapes use 16 / 8), ragged channel block (80 = 64 + 16), several channel blocks each
    way, wrap-around of the odd-row ring"""
    monkeypatch.setenv("CSEG_S2_WRW_RPU", rpu)
    B, ci, co, Ho, Wo = case
    x, dy = _rand((B, ci, 2 * Ho, 2 * Wo), 75, 7.0), _rand((B, co, Ho, Wo), 76, 1e-4)
    dw = E.conv3x3_s2_wrw(x, dy)
    ref = E.ref_conv3x3_s2_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * Ho * Wo) * float(np.abs(ref).max())
    assert np.array_equal(dw, E.conv3x3_s2_wrw(x, dy))          # fixed-order reduction: bit-identical on a second run


# ---- the first stem convolution, nn.Conv2d(3, 64, 3, 2, 1) on the image (csrc/conv3x3_stem.hip, round 6): fp32 FMA kernels -------
RGB_CASES = [(1, 8, 128), (2, 10, 132), (1, 2, 6), (3, 18, 260)]      # B, H, W of the image: one tile, ragged tiles both ways, smaller than a tile, several


@pytest.mark.parametrize("case", RGB_CASES)
def test_rgb_stem_forward_and_weight_gradient(case, wave_order):
    B, H, W = case
    x, w = _rand((B, 3, H, W), 91, 2.0), _rand((64, 3, 3, 3), 92, 0.2)
    y = E.conv3x3_s2_rgb(x, w)
    ref = E.ref_conv3x3_s2(x, w)
    assert not np.isnan(y).any()                                       # every output element written (NaN-filled buffer)
    assert np.abs(y - ref).max() <= 4e-6 * float(np.abs(ref).max())    # fp32 multiply-adds, K = 27
    dy = _rand((B, 64, H // 2, W // 2), 93, 1e-3)
    dw = E.conv3x3_s2_rgb_wrw(x, dy)
    ref = E.ref_conv3x3_s2_wrw(x, dy)
    assert not np.isnan(dw).any()
    assert np.abs(dw - ref).max() <= 2e-6 * np.sqrt(B * H * W / 4) * float(np.abs(ref).max())
    assert np.array_equal(dw, E.conv3x3_s2_rgb_wrw(x, dy))             # fixed-order sums: bit-identical on a second run


# ---- transition 1 of HRNet (256 -> 48 at stride 1, 256 -> 96 at stride 2): channel counts that are multiples of 64, not of 48 ------
def test_transition_layer_256_to_48(wave_order):
    """forward (conv_out 48, 16 chunks streamed), backward-data (conv_out 256: four channel tiles per block, 16-channel-chunk
    kernel), weight gradient with different channel counts (four 64-wide input blocks)"""
    B, ci, co, H, W = 1, 256, 48, 5, 64
    x, w, dy = _rand((B, ci, H, W), 81, 2.0), _rand((co, ci, 3, 3), 82, 0.05), _rand((B, co, H, W), 83, 1e-3)
    y = E.conv3x3_sb(x, w, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w)
    assert not np.isnan(y).any() and np.abs(y - ref).max() <= _bound(ref, 9 * ci)
    dx = E.conv3x3_sb(dy, w, transpose_flip=True, arith=E.F16X3)
    ref = E.ref_conv3x3_bwd_data(dy, w)
    assert dx.shape == (B, ci, H, W) and not np.isnan(dx).any()
    assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(9 * co) * float(np.abs(ref).max())
    dw = E.conv3x3_sb_wrw(x, dy, arith=E.F16X3)
    ref = E.ref_conv3x3_wrw(x, dy)
    assert not np.isnan(dw).any() and np.abs(dw - ref).max() <= 3e-6 * np.sqrt(B * H * W) * float(np.abs(ref).max())


def test_stride2_backward_data_256_input_channels(wave_order):
    """256 -> 96 at stride 2: dx has 256 channels = four tiles per block"""
    B, ci, co, Ho, Wo = 1, 256, 96, 3, 34
    dy, w = _rand((B, co, Ho, Wo), 84, 1e-3), _rand((co, ci, 3, 3), 85, 0.05)
    dx = E.conv3x3_s2_bwd(dy, w, 4)
    ref = E.ref_conv3x3_s2_bwd_data(dy, w)
    assert not np.isnan(dx).any()
    assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(4 * co) * float(np.abs(ref).max())


# ---- the 8 x 64-pixel head kernel (conv3x3_sb16.hip, namespace sb8; nt = CSEG_NT_SB8) ----------------------------------------------
NT_SB8 = 0x109


@pytest.mark.parametrize("case", [(1, 48, 144, 9, 68), (2, 16, 144, 8, 64), (1, 32, 288, 3, 20)])
def test_head_kernel_8_rows(case, wave_order):
    """ragged tiles both ways (9 rows = one full + one single-row tile, 68 columns), one / two / three 16-channel chunks (raw patch of
    chunk c + 1 brought in by LDS-DMA in five parts during the K-steps of chunk c), two channel tile groups, bias, and the
    backward-data operator (mirrored packing) of the same weights"""
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 91, 3.0), _rand((co, ci, 3, 3), 92, 0.1), _rand((co,), 93)
    y = E.conv3x3_sb(x, w, bias=b, nt=NT_SB8, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w, b)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)
    # backward-data: the operator maps co -> ci channels, so it needs ci % 144 == 0: use square weights
    w2 = _rand((144, 144, 3, 3), 94, 0.05)
    dy = _rand((1, 144, 5, 36), 95, 1e-3)
    dx = E.conv3x3_sb(dy, w2, transpose_flip=True, nt=NT_SB8, arith=E.F16X3)
    ref = E.ref_conv3x3_bwd_data(dy, w2)
    assert not np.isnan(dx).any()
    assert np.abs(dx - ref).max() <= 3e-6 * np.sqrt(9 * 144) * float(np.abs(ref).max())


@pytest.mark.parametrize("case", [(1, 48, 144, 9, 68), (2, 32, 288, 8, 64)])
def test_head_kernel_version_2_is_bit_identical_to_version_1(case, wave_order, monkeypatch, tmp_path):
    """conv3x3_sb8p_kernel (round 6: weight stages as a ring of three, the next K-step's fragments read before the barrier, raw-patch
    line offsets computed once) against conv3x3_sb8_kernel (CSEG_SB8_V=1): same packed weights, K-steps and accumulation order, so
    the outputs must agree to the last bit -- three / two chunks (the read-ahead crosses K-steps but never a chunk boundary)."""
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 191, 3.0), _rand((co, ci, 3, 3), 192, 0.1), _rand((co,), 193)
    trace = tmp_path / "launches.txt"
    monkeypatch.setenv("CSEG_EMU_TRACE", str(trace))
    y2 = E.conv3x3_sb(x, w, bias=b, nt=NT_SB8, arith=E.F16X3)
    assert "conv3x3_sb8p_kernel<" in trace.read_text()
    monkeypatch.setenv("CSEG_SB8_V", "1")
    y1 = E.conv3x3_sb(x, w, bias=b, nt=NT_SB8, arith=E.F16X3)
    assert "conv3x3_sb8_kernel<" in trace.read_text()
    assert not np.isnan(y2).any()
    assert np.array_equal(y1, y2)


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("case", [(1, 48, 48, 6, 68), (2, 16, 96, 4, 64), (1, 64, 64, 5, 20), (1, 32, 192, 3, 36)])
def test_epilogue_addend(case, arith, wave_order):
    """y = conv(x) + bias + addend in the three forward kernels that take branch layers (persistent 16-channel-chunk kernel at 48 / 192
    channels, its one-tile form at 64, the 32-channel-chunk kernel at 96), ragged tiles: full float4 and scalar edge stores"""
    B, ci, co, H, W = case
    x, w, b, ad = _rand((B, ci, H, W), 101, 2.0), _rand((co, ci, 3, 3), 102, 0.1), _rand((co,), 103), _rand((B, co, H, W), 104, 5.0)
    y = E.conv3x3_sb(x, w, bias=b, arith=arith, addend=ad)
    ref = E.ref_conv3x3(x, w, b) + ad
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() <= _bound(ref, 9 * ci)


# ---- round 4 (CSEG_SB16_ROWS8; default where the tiles fill 256 blocks): 8 x 64-pixel tiles, one wave per output row x three channel tiles (conv3x3_sb16r_kernel) ----
@pytest.mark.parametrize("case,what", [
    ((1, 48, 48, 11, 68), "weights resident + ONE patch buffer (two barriers per chunk), ragged tiles both ways (11 rows, 68 columns)"),
    ((2, 16, 48, 8, 64), "one chunk per tile, exact tiles"),
    ((1, 32, 48, 208, 640), "260 tiles on 256 blocks: some blocks walk two tiles with the weights resident"),
    ((3, 16, 48, 100, 640), "390 tiles: XCD ranges of 49 tiles, the last one short (47), blocks with one and with two tiles"),
    ((1, 96, 48, 9, 68), "weights streamed (6 chunks do not fit), two patch buffers, ragged tiles"),
    ((2, 96, 96, 104, 640), "two channel tile groups, 260 tiles each on 128 blocks: streamed weights across tile boundaries"),
])
def test_eight_row_tiles_are_bit_identical_to_the_four_row_kernels(case, what, monkeypatch, tmp_path):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", "shuffle:5")
    monkeypatch.setenv("CSEG_CONV3X3_SB16_CH", "48,96")            # 96 output channels on the 16-channel-chunk kernels too (3 tiles per block)
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 91, 2.0), _rand((co, ci, 3, 3), 92, 1.0 / (3 * ci ** 0.5)), _rand((co,), 93)
    monkeypatch.setenv("CSEG_SB16_ROWS8", "0")                     # the 4-row kernels
    y0 = E.conv3x3_sb(x, w, b, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w, b)
    assert np.abs(y0 - ref).max() <= _bound(ref, 9 * ci), what
    monkeypatch.setenv("CSEG_SB16_ROWS8", "2")                     # 2: also where the 8-row tiles do not fill 256 blocks
    trace = tmp_path / "launches.txt"
    monkeypatch.setenv("CSEG_EMU_TRACE", str(trace))
    y1 = E.conv3x3_sb(x, w, b, arith=E.F16X3)
    assert "conv3x3_sb16r_kernel" in trace.read_text(), "the switch did not route to the 8-row kernel"
    assert not np.isnan(y1).any()
    assert np.array_equal(y0, y1), "same packed weights, same K-steps, same accumulation order per output element (%s)" % what
    monkeypatch.setenv("CSEG_SB16_XCD", "1")                       # XCD-contiguous tile order (one channel tile group only): every tile once
    y2 = E.conv3x3_sb(x, w, b, arith=E.F16X3)
    assert np.array_equal(y0, y2), "the XCD-aware tile order must compute every tile exactly once (%s)" % what


# ---- round 4 (opt-in, CSEG_XCD_REMAP): XCD-aware block order of the one-tile kernels -- a permutation of the grid, same results ----
@pytest.mark.parametrize("case,nt,kernel,what", [
    ((2, 32, 96, 8, 128), 0, "conv3x3_sb_kernel", "6 channel tiles: 2 images x 2 row tiles x 2 column tiles = 8 blocks"),
    ((1, 32, 64, 30, 72), 0, "conv3x3_sb16_kernel", "4 channel tiles: 8 row tiles x 2 column tiles = 16 blocks, ragged"),
    ((2, 16, 288, 9, 68), NT_SB8, "conv3x3_sb8p_kernel", "head kernel: 2 images x 2 channel tile groups x 2 x 2 tiles = 16 blocks, ragged"),
    ((1, 32, 96, 12, 64), 0, "conv3x3_sb_kernel", "3 blocks: not a multiple of 8, the remap must be the identity"),
])
def test_xcd_block_order_is_a_permutation(case, nt, kernel, what, monkeypatch, tmp_path):
    monkeypatch.setenv("CSEG_EMU_WAVE_ORDER", "shuffle:3")
    monkeypatch.setenv("CSEG_CONV3X3_SB16_P", "0")                 # the one-tile form of the 16-channel-chunk kernel
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 101, 2.0), _rand((co, ci, 3, 3), 102, 1.0 / (3 * ci ** 0.5)), _rand((co,), 103)
    y0 = E.conv3x3_sb(x, w, bias=b, nt=nt, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w, b)
    assert np.abs(y0 - ref).max() <= _bound(ref, 9 * ci), what
    monkeypatch.setenv("CSEG_XCD_REMAP", "1")
    trace = tmp_path / "launches.txt"
    monkeypatch.setenv("CSEG_EMU_TRACE", str(trace))
    y1 = E.conv3x3_sb(x, w, bias=b, nt=nt, arith=E.F16X3)
    launched = trace.read_text()
    assert kernel + "<" in launched, "expected %s, got %s" % (kernel, launched)
    assert np.array_equal(y0, y1), what


@pytest.mark.parametrize("case", [(2, 64, 288, 8, 64), (1, 48, 96, 5, 20)])
def test_xcd_block_order_of_the_pointwise_kernel(case, monkeypatch, tmp_path):
    """1x1 convolution under CSEG_XCD_REMAP=1 (channel tile group fastest inside contiguous per-XCD runs): a permutation of the grid --
    2 images x 2 pixel tiles x 2 channel tile groups = 8 blocks in the first case, the identity in the second (one block)."""
    B, ci, co, H, W = case
    x, w, b = _rand((B, ci, H, W), 111, 2.0), _rand((co, ci, 1, 1), 112, 1.0 / ci ** 0.5), _rand((co,), 113)
    y0 = E.conv1x1_sb(x, w, b, arith=E.F16X3)
    monkeypatch.setenv("CSEG_XCD_REMAP", "1")
    trace = tmp_path / "launches.txt"
    monkeypatch.setenv("CSEG_EMU_TRACE", str(trace))
    y1 = E.conv1x1_sb(x, w, b, arith=E.F16X3)
    assert "conv1x1_sb_kernel<" in trace.read_text()
    assert np.array_equal(y0, y1)


@pytest.mark.parametrize("case", [(1, 16, 48, 128, 256), (1, 16, 48, 72, 256), (1, 32, 192, 32, 64), (2, 32, 192, 32, 64), (3, 16, 384, 16, 32),
                                  (1, 16, 96, 64, 128)])
def test_xcd_tile_order_of_the_persistent_kernels_is_a_permutation(case, monkeypatch):
    """CSEG_SB16_XCD (default 1) against the plain tile order at the tile counts of small per-GPU batches: 128 and 72 tiles on as many
    blocks (one channel tile group), 8 / 16 / 12 tiles x 4 / 8 channel tile groups (the n_cot blocks of a tile on one XCD), 2 groups."""
    monkeypatch.setenv("CSEG_CONV3X3_SB16_CH", "48,96,192,384")
    B, ci, co, H, W = case
    x, w = _rand((B, ci, H, W), 121, 2.0), _rand((co, ci, 3, 3), 122, 1.0 / (3 * ci ** 0.5))
    monkeypatch.setenv("CSEG_SB16_XCD", "0")
    y0 = E.conv3x3_sb(x, w, None, arith=E.F16X3)
    monkeypatch.setenv("CSEG_SB16_XCD", "1")
    y1 = E.conv3x3_sb(x, w, None, arith=E.F16X3)
    ref = E.ref_conv3x3(x, w, None)
    assert np.abs(y1 - ref).max() <= _bound(ref, 9 * ci)
    assert np.array_equal(y0, y1)
