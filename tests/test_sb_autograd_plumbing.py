"""Autograd plumbing of the split-bf16 convolution Functions (contrastiveseg_amd/kernels.py) on CPU: the device entry
points are replaced by plain torch convolutions, so what is tested is the host logic -- which gradient comes from which
call, argument order, bias handling, the switches -- against ordinary autograd. (The kernels themselves: -m gpu.)"""
import pytest
import torch
import torch.nn.functional as F

from contrastiveseg_amd import kernels as K


@pytest.fixture
def cpu_entry_points(monkeypatch):
    calls = []

    def sb3(x, w, transpose_flip=False, bias=None, nt=0, ax=None, addend=None, want_stats=False):
        calls.append(("sb3", bool(transpose_flip), bias is not None, nt))
        return F.conv_transpose2d(x, w, None, 1, 1) if transpose_flip else F.conv2d(x, w, bias, 1, 1)

    def wrw3(x, dy, ax=None, ady=None):
        calls.append(("wrw3",))
        return torch.nn.grad.conv2d_weight(x, (dy.shape[1], x.shape[1], 3, 3), dy, padding=1)

    def sb1(x, w, transpose=False, bias=None, ax=None, want_stats=False):
        calls.append(("sb1", bool(transpose), bias is not None))
        return F.conv_transpose2d(x, w) if transpose else F.conv2d(x, w, bias)

    def wrw1(x, dy, ax=None, ady=None):
        calls.append(("wrw1",))
        return torch.nn.grad.conv2d_weight(x, (dy.shape[1], x.shape[1], 1, 1), dy)

    monkeypatch.setattr(K, "tensor_amax", lambda t, slot=None: None)      # the device word of max|t| (f16x3 arithmetic)
    monkeypatch.setattr(K, "conv3x3_sb_run", sb3)
    monkeypatch.setattr(K, "_conv3x3_wrw", lambda x, dy, co, ci: wrw3(x, dy))
    monkeypatch.setattr(K, "conv3x3_sb_wrw", wrw3)
    monkeypatch.setattr(K, "conv3x3_sb_wrw_eligible", lambda x, dy: True)
    monkeypatch.setattr(K, "conv1x1_sb_run", sb1)
    monkeypatch.setattr(K, "conv1x1_sb_wrw", wrw1)
    monkeypatch.setattr(K, "conv1x1_sb_wrw_eligible", lambda x, dy: True)
    return calls


def _grads(fn, x, w, b, dy):
    xs, ws = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bs = None if b is None else b.clone().requires_grad_(True)
    fn(xs, ws, bs).backward(dy)
    return xs.grad, ws.grad, None if bs is None else bs.grad


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("sb_wrw", [False, True])
@pytest.mark.parametrize("channels", [48, 192])
def test_conv3x3_function(cpu_entry_points, monkeypatch, bias, sb_wrw, channels):
    monkeypatch.setattr(K, "CONV3X3_SB_WRW", sb_wrw)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, channels, 6, 8, generator=g)
    w = torch.randn(channels, channels, 3, 3, generator=g) / 20
    b = torch.randn(channels, generator=g) if bias else None
    dy = torch.randn(2, channels, 6, 8, generator=g)
    got = _grads(K.conv3x3_split_bf16, x, w, b, dy)
    want = _grads(lambda a, c, d: F.conv2d(a, c, d, 1, 1), x, w, b, dy)
    for a, e in zip(got, want):
        assert (a is None) == (e is None)
        if a is not None:
            assert torch.allclose(a, e, rtol=1e-4, atol=1e-4)
    kinds = [c[0] for c in cpu_entry_points]
    assert kinds.count("sb3") == 2                               # forward + backward-data
    fwd, bwd = [c for c in cpu_entry_points if c[0] == "sb3"]
    assert fwd[1] is False and fwd[2] is bias and bwd[1] is True and bwd[2] is False
    expect_nt = K.conv3x3_sb_pick_nt(x, channels) if channels in K.CONV3X3_SB_PICK_NT_CHANNELS else 0
    assert fwd[3] == expect_nt and bwd[3] == expect_nt
    # weight gradient: split kernel when switched on AND the channel count is one it was timed on (48 / 96 / 720), else the
    # fp32-MFMA kernel for the bias-free 48/96 branches, else aten
    uses_custom_wrw = (sb_wrw and channels in K.CONV3X3_SB_WRW_CHANNELS) or (not bias and channels in K.CONV3X3_WRW_CHANNELS)
    assert ("wrw3" in kinds) == uses_custom_wrw


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("sb_wrw", [False, True])
def test_conv1x1_function(cpu_entry_points, monkeypatch, bias, sb_wrw):
    monkeypatch.setattr(K, "CONV1X1_SB_WRW", sb_wrw)
    monkeypatch.setattr(K, "CONV1X1_SB_WRW_MIN_CH", 16)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 48, 4, 8, generator=g)
    w = torch.randn(64, 48, 1, 1, generator=g) / 7
    b = torch.randn(64, generator=g) if bias else None
    dy = torch.randn(2, 64, 4, 8, generator=g)
    got = _grads(K.conv1x1_split_bf16, x, w, b, dy)
    want = _grads(lambda a, c, d: F.conv2d(a, c, d), x, w, b, dy)
    for a, e in zip(got, want):
        assert (a is None) == (e is None)
        if a is not None:
            assert torch.allclose(a, e, rtol=1e-4, atol=1e-4)
    kinds = [c[0] for c in cpu_entry_points]
    assert kinds.count("sb1") == 2 and ("wrw1" in kinds) == sb_wrw


def test_frozen_input_skips_backward_data(cpu_entry_points):
    x = torch.randn(1, 48, 4, 4)                                 # no grad: e.g. the first layer after the image
    w = torch.randn(48, 48, 3, 3, requires_grad=True)
    K.conv3x3_split_bf16(x, w, None).sum().backward()
    assert [c[0] for c in cpu_entry_points].count("sb3") == 1 and w.grad is not None
