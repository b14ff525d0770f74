"""Thin autograd wrappers over the C-ABI (contrastiveseg_amd/_hip.py). Same pattern as the reference's own native
ops (lib/extensions/cc_attention/functions.py:20-47): a torch.autograd.Function that pre-allocates outputs, passes
raw pointers + the current stream, and raises RuntimeError on a 0 return. No CPU path."""
import ctypes

import os

import torch
from torch.autograd import Function

from . import _hip

I32, I16, I64, F32 = torch.int32, torch.int16, torch.int64, torch.float32


def _p(t, dtype, what):
    return _hip.dev(t, dtype, what)


def _null():
    return ctypes.c_void_p(None)


def _on_device(t):
    """Lives where the library's kernels can read it (the GPU). The emulated-device tests (tests/emu/inject.py) replace
    this together with _hip.dev; the product itself has no CPU path."""
    return t.is_cuda


def _pq(t, what):
    """Raw pointer of an fp32 operand of a hot entry point: one combined test on the fast path (device, dtype, layout), the full
    _hip.dev() diagnosis only when it fails. (1.2 -> 0.5 us per operand, ~3 000 operands per batch-8 step.)"""
    if t.dtype == F32 and t.is_contiguous() and _on_device(t) and (not t.is_cuda or t.device.index == _hip._raw_device()):
        return ctypes.c_void_p(t.data_ptr())
    return _hip.dev(t, F32, what)


def _pf(t):
    """Raw pointer of a tensor this module has just allocated itself (device, dtype and layout known by construction):
    skips the four checks of _hip.dev(). The step issues ~1000 calls into the BN / convolution entry points and is
    host-launch-bound at small per-GPU batches; user-facing tensors (x, dy, residual, weights) are still validated."""
    return ctypes.c_void_p(t.data_ptr())


# Host cost: the functions of this module that only allocate outputs and call the library (the BN and convolution entry points) carry
# no @torch.no_grad(): they are reached from inside autograd Functions, where recording is off already, they use no differentiable torch
# op, and the decorator's context object was 4 ms of host time per batch-8 step (3 400 entries, tools/host_profile.py, round 4).
# ----------------------------------------------------------------------------------------------------------
# anchor mining
# ----------------------------------------------------------------------------------------------------------
@torch.no_grad()
def classify_partition(target, ignore_label, seg=None, predict=None, num_classes=None, feat_hw=None,
                       want_maps=False):
    """lib/loss/loss_contrast.py:131-134 + :183 + the unique/nonzero bookkeeping of :35-64, on the device.
    Returns dict(counts [B,K,2], seg_off [B,K,2], part_idx [B,P], status [4], lab/pred [B,P] if want_maps)."""
    B, H, W = target.shape
    target = target.contiguous()
    if seg is not None:
        seg = seg.contiguous()          # NCHW planes are what the kernel walks (channels_last models included)
        _, K, h, w = seg.shape
        seg_p, pred_p = _p(seg, F32, "seg"), _null()
    else:
        K = int(num_classes)
        h, w = feat_hw
        predict = predict.reshape(B, h * w)
        seg_p, pred_p = _null(), _p(predict, I64, "predict")
    P = h * w
    dev = target.device
    out = {
        "counts": torch.empty(B, K, 2, dtype=I32, device=dev),
        "seg_off": torch.empty(B, K, 2, dtype=I32, device=dev),
        "part_idx": torch.empty(B, P, dtype=I32, device=dev),
        "status": torch.empty(4, dtype=I32, device=dev),
        "key": torch.empty(B, P, dtype=I16, device=dev),
    }
    if want_maps:
        out["lab"] = torch.empty(B, P, dtype=I32, device=dev)
        out["pred"] = torch.empty(B, P, dtype=I32, device=dev)
    _hip.call("cseg_classify_partition", seg_p, pred_p, _p(target, I64, "target"), B, K, h, w, H, W,
              int(ignore_label),
              _p(out["lab"], I32, "lab") if want_maps else _null(),
              _p(out["pred"], I32, "pred") if want_maps else _null(),
              _p(out["key"], I16, "key"), _p(out["counts"], I32, "counts"), _p(out["seg_off"], I32, "seg_off"),
              _p(out["part_idx"], I32, "part_idx"), _p(out["status"], I32, "status"), _hip.stream_ptr())
    return out


@torch.no_grad()
def gather_anchors(embed, part_idx, sel_pos):
    B, D = embed.shape[:2]
    P = embed.shape[2] * embed.shape[3]
    N = sel_pos.numel()
    anchors = torch.empty(N, D, dtype=F32, device=embed.device)
    sel_pix = torch.empty(N, dtype=I32, device=embed.device)
    _hip.call("cseg_gather_anchors", _p(embed, F32, "embed"), B, D, P, _p(part_idx, I32, "part_idx"),
              _p(sel_pos, I32, "sel_pos"), N, _p(anchors, F32, "anchors"), _p(sel_pix, I32, "sel_pix"),
              _hip.stream_ptr())
    return anchors, sel_pix


# ----------------------------------------------------------------------------------------------------------
# contrastive term
# ----------------------------------------------------------------------------------------------------------
def _desc(mode, anchors, a_lab, temperature, base_temperature, contrast=None, c_lab=None, segment_queue=None,
          pixel_queue=None):
    d = _hip.ContrastDesc()
    d.mode = mode
    d.N, d.D = anchors.shape
    d.anchors = _p(anchors, F32, "anchors")
    d.a_lab = _p(a_lab, I32, "a_lab")
    d.temperature = float(temperature)
    d.base_temperature = float(base_temperature)
    if mode == 0:
        d.M = d.N
    elif mode == 1:
        d.M = contrast.shape[0]
        d.contrast = _p(contrast, F32, "contrast")
        d.c_lab = _p(c_lab, I32, "c_lab")
    else:
        K, ms, Dq = segment_queue.shape
        if Dq != d.D or tuple(pixel_queue.shape) != (K, ms, Dq):
            raise RuntimeError("queue shapes %s / %s do not match D=%d" % (tuple(segment_queue.shape),
                                                                           tuple(pixel_queue.shape), d.D))
        d.M = K * 2 * ms
        d.segment_queue = _p(segment_queue, F32, "segment_queue")
        d.pixel_queue = _p(pixel_queue, F32, "pixel_queue")
        d.bank_classes, d.bank_size = K, ms
    return d


# Forward of the contrastive term (csrc/contrast.hip): "0" = three launches (S = A.C^T/tau to HBM, row pass over S, mean), "1" = ONE
# launch (round 5: S tiles on the fp32 MFMA, online row statistics by wavefront reductions, the positives' sweep from LDS after an
# in-launch hand-off between the blocks of a row strip) that also stores the S tiles once for the backward. Both are kept: the default
# is whichever measured faster on the MI355X (bench_detail.json carries both timings).
# Measured on the MI355X (profiles/r05_contrast_fused_probe.json, host + device time of back-to-back calls): self N = 912 37.0 us (three)
# vs 52.4 (fused); bank 1024 x 4104 56.7 vs 105.9; bank 152 x 190 000 (the reference's memory_size 5000) 936 vs 670. The single launch
# pays two in-launch hand-offs and a serial finish (last block of a strip -> last strip -> mean) that three back-to-back launches do not,
# and wins once S is large enough for its HBM round trips to matter. "auto" (default): fused from N * M >= 2^24 on.
CONTRAST_FUSED = os.environ.get("CSEG_CONTRAST_FUSED", "auto")
CONTRAST_FUSED_MIN_NM = 1 << 24
_FUSED_WS = {}       # (device index, stream) -> scratch of the fused forward; its counters are zero between launches (the kernel resets them)


def _fused_ws(N, M, device):
    lib = _hip.lib()
    need = lib.cseg_contrast_fused_ws_bytes(N, M) // 4
    key = (device.index, _hip.raw_stream()) if device.type == "cuda" else (-1, 0)
    buf = _FUSED_WS.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros(max(need, 1 << 16), dtype=F32, device=device)         # zero ONCE: the counters live inside
        _FUSED_WS[key] = buf
    return buf


def contrast_forward(desc, device):
    """Runs cseg_contrast_fwd / cseg_contrast_fwd_fused. Returns (loss [1], saved) where saved feeds contrast_backward."""
    lib = _hip.lib()
    ws = lib.cseg_contrast_ws_bytes(desc.N, desc.M)
    S = torch.empty(ws // 4, dtype=F32, device=device)
    row_stats = torch.empty(desc.N, 4, dtype=F32, device=device)
    row_loss = torch.empty(desc.N, dtype=F32, device=device)
    loss = torch.empty(1, dtype=F32, device=device)
    if CONTRAST_FUSED == "1" or (CONTRAST_FUSED == "auto" and desc.N * desc.M >= CONTRAST_FUSED_MIN_NM):
        # (the scratch is laid out by the library for exactly this (N, M): counters behind the partials -- a buffer that served another
        # shape still has its counters at zero, wherever they were)
        scratch = _fused_ws(desc.N, desc.M, device)
        off = lib.cseg_contrast_fused_counter_offset(desc.N, desc.M) // 4
        n_ctr = lib.cseg_contrast_fused_ws_bytes(desc.N, desc.M) // 4 - off
        key = (off, n_ctr)                                # where the counters of THIS launch plan sit
        if getattr(scratch, "_cseg_shape", key) != key:
            scratch[off:off + n_ctr].zero_()              # another (N, M) left ITS counters zero, but partials may sit where ours are
        scratch._cseg_shape = key
        _hip.call("cseg_contrast_fwd_fused", ctypes.byref(desc), _pf(scratch), _p(S, F32, "S_out"), _p(row_stats, F32, "row_stats"),
                  _p(row_loss, F32, "row_loss"), _p(loss, F32, "loss"), _hip.stream_ptr())
        return loss, (S, row_stats, row_loss)
    _hip.call("cseg_contrast_fwd", ctypes.byref(desc), _p(S, F32, "S_ws"), _p(row_stats, F32, "row_stats"),
              _p(row_loss, F32, "row_loss"), _p(loss, F32, "loss"), _hip.stream_ptr())
    return loss, (S, row_stats, row_loss)


def contrast_backward(desc, saved, d_loss, device):
    """Runs cseg_contrast_bwd. Returns d_anchor_parts [n_parts, N, D]."""
    lib = _hip.lib()
    S, row_stats, _ = saved
    n_parts = lib.cseg_contrast_bwd_parts(desc.N, desc.M, desc.D)
    parts = torch.empty(n_parts, desc.N, desc.D, dtype=F32, device=device)
    d_loss = d_loss.reshape(1).to(F32).contiguous()
    _hip.call("cseg_contrast_bwd", ctypes.byref(desc), _p(S, F32, "S_ws"), _p(row_stats, F32, "row_stats"),
              _p(d_loss, F32, "d_loss"), _p(parts, F32, "d_anchor_parts"), _hip.stream_ptr())
    return parts


def _bank_versions(segment_queue, pixel_queue):
    return None if segment_queue is None else (segment_queue._version, pixel_queue._version)


def _check_bank_unchanged(versions, segment_queue, pixel_queue):
    """The bank is read IN PLACE by cseg_contrast_fwd and again by cseg_contrast_bwd (no [K*2*ms, D] copy like the
    reference's torch.cat, loss_contrast_mem.py:221). Raw pointers bypass autograd's own version check, so it is
    restated here: an in-place update of the queues between forward and backward would silently mix old similarities
    with new bank rows."""
    if versions is not None and versions != (segment_queue._version, pixel_queue._version):
        raise RuntimeError("the memory bank was modified in place between the forward and the backward of the "
                           "contrastive loss (call _dequeue_and_enqueue AFTER loss.backward(), as "
                           "Trainer.train_step does, or pass clones of the queues)")


# Row-sparse hand-over of the embedding gradient (default since round 3: parity "ok" and 166.9 -> 163.3 ms/step in the round-2
# driver pass, GPUTEST_r02.json; CSEG_SPARSE_EMBED_GRAD=0 restores the dense route): the contrastive term touches
# <= max_samples pixels of the [B,D,h,w] embedding, so its gradient is N rows and the rest zeros. With the switch on, the
# projection head (lib/models/modules/projection.py) tags the embedding it returns with a SparseGradSlot; the loss'
# backward deposits (rows, pixel indices) there and returns a zero tensor WITHOUT storage (stride 0) as the dense
# gradient, and the head's backward works on the N rows only (no 268 MB memset + scatter, no dense normalise / 1x1
# backward at bs 8). Any other consumer of the embedding simply adds a real dense gradient, which the head detects.
SPARSE_EMBED_GRAD = os.environ.get("CSEG_SPARSE_EMBED_GRAD", "1") == "1"


class SparseGradSlot(object):
    __slots__ = ("deposits", "_sentinel")

    def __init__(self):
        self.deposits = []          # [(rows [n,D] f32, sel_pix [n] i32 = b * P + pixel)]
        self._sentinel = None

    def deposit(self, rows, sel_pix, shape):
        """Stores the rows and returns the storage-free stand-in for the dense gradient of an embedding of `shape`."""
        self.deposits.append((rows, sel_pix))
        self._sentinel = rows.new_zeros((1,) * len(shape)).expand(shape)
        return self._sentinel

    def is_standin(self, g):
        """`g` is the storage-free zero tensor deposit() handed out. Strides of size-1 dimensions are not compared: expanding a
        [1,1,1,1] tensor to a batch of ONE image leaves that dimension's stride at 1 (round 3 tested `not any(g.stride())`, so a
        per-GPU batch of one image -- the 8-GPU strong-scaling point -- silently took the dense route)."""
        s = self._sentinel
        return (s is not None and g.data_ptr() == s.data_ptr()
                and all(st == 0 for st, n in zip(g.stride(), g.shape) if n > 1))

    def take(self):
        d, self.deposits, self._sentinel = self.deposits, [], None
        return d


class ContrastOnAnchors(Function):
    """loss = _contrastive(anchors, ...) with anchors already gathered ([N,D], view-major rows).
    mode: 'self' | 'plain' | 'bank'. Gradient flows to `anchors` only (the reference gives the bank none,
    loss_contrast_mem.py:221 reads buffers). Used by tests and by the cross-rank contrast set."""

    @staticmethod
    def forward(ctx, anchors, a_lab, mode, temperature, base_temperature, contrast, c_lab, segment_queue,
                pixel_queue):
        anchors = anchors.contiguous()
        m = {"self": 0, "plain": 1, "bank": 2}[mode]
        desc = _desc(m, anchors, a_lab, temperature, base_temperature, contrast, c_lab, segment_queue, pixel_queue)
        loss, saved = contrast_forward(desc, anchors.device)
        ctx.desc = desc
        ctx.saved = saved
        ctx.keep = (anchors, a_lab, contrast, c_lab, segment_queue, pixel_queue)  # keep pointers alive
        ctx.bank_versions = _bank_versions(segment_queue, pixel_queue)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        anchors = ctx.keep[0]
        _check_bank_unchanged(ctx.bank_versions, ctx.keep[4], ctx.keep[5])
        parts = contrast_backward(ctx.desc, ctx.saved, g, anchors.device)
        return (parts.sum(0) if parts.shape[0] > 1 else parts[0],) + (None,) * 8


class PixelContrast(Function):
    """gather(embed NCHW at mined pixels) -> _contrastive -> scalar, fused end to end; backward = contrast
    backward + scatter of the anchor rows into a zero d_embed (no NHWC copy, no dense N x N temporaries besides the
    similarity workspace). mode: 'self' (contrast set = the anchors) | 'bank' (memory queues read in place)."""

    @staticmethod
    def forward(ctx, embed, part_idx, sel_pos, a_lab, mode, temperature, base_temperature, segment_queue,
                pixel_queue, slot=None):
        if not embed.is_contiguous():
            embed = embed.contiguous()
        anchors, sel_pix = gather_anchors(embed, part_idx, sel_pos)
        ctx.slot = slot
        m = {"self": 0, "bank": 2}[mode]
        desc = _desc(m, anchors, a_lab, temperature, base_temperature, None, None, segment_queue, pixel_queue)
        loss, saved = contrast_forward(desc, embed.device)
        ctx.desc, ctx.saved = desc, saved
        ctx.keep = (anchors, a_lab, segment_queue, pixel_queue)
        ctx.bank_versions = _bank_versions(segment_queue, pixel_queue)
        ctx.sel_pix = sel_pix
        ctx.embed_shape = embed.shape
        ctx.mark_non_differentiable(sel_pix)
        return loss.reshape(()), sel_pix

    @staticmethod
    def backward(ctx, g, _g_sel):
        B, D, h, w = ctx.embed_shape
        dev = ctx.sel_pix.device
        _check_bank_unchanged(ctx.bank_versions, ctx.keep[2], ctx.keep[3])
        parts = contrast_backward(ctx.desc, ctx.saved, g, dev)
        if ctx.slot is not None:
            rows = parts.sum(0) if parts.shape[0] > 1 else parts[0]
            return (ctx.slot.deposit(rows, ctx.sel_pix, ctx.embed_shape),) + (None,) * 9
        d_embed = torch.zeros(B, D, h, w, dtype=F32, device=dev)
        _hip.call("cseg_scatter_anchor_grad", _p(parts, F32, "parts"), parts.shape[0], _p(ctx.sel_pix, I32, "sel_pix"),
                  parts.shape[1], D, h * w, 1.0, _p(d_embed, F32, "d_embed"), _hip.stream_ptr())
        return (d_embed,) + (None,) * 9


class GatherAnchors(Function):
    """anchors = embed[b, :, pix] rows (differentiable): used when the contrast set is assembled across ranks."""

    @staticmethod
    def forward(ctx, embed, part_idx, sel_pos, slot=None):
        if not embed.is_contiguous():
            embed = embed.contiguous()
        anchors, sel_pix = gather_anchors(embed, part_idx, sel_pos)
        ctx.sel_pix = sel_pix
        ctx.embed_shape = embed.shape
        ctx.slot = slot
        ctx.mark_non_differentiable(sel_pix)
        return anchors, sel_pix

    @staticmethod
    def backward(ctx, g, _):
        B, D, h, w = ctx.embed_shape
        g = g.contiguous()
        if ctx.slot is not None:
            return ctx.slot.deposit(g, ctx.sel_pix, ctx.embed_shape), None, None, None
        d_embed = torch.zeros(B, D, h, w, dtype=F32, device=g.device)
        _hip.call("cseg_scatter_anchor_grad", _p(g, F32, "d_anchors"), 1, _p(ctx.sel_pix, I32, "sel_pix"),
                  g.shape[0], D, h * w, 1.0, _p(d_embed, F32, "d_embed"), _hip.stream_ptr())
        return d_embed, None, None, None


# ----------------------------------------------------------------------------------------------------------
# HRNet head: upsample + concat
# ----------------------------------------------------------------------------------------------------------
def _int_arr(vals):
    return (ctypes.c_int * len(vals))(*vals)


# max|.| records from the kernels that WRITE a tensor read by split-operand convolutions (upsample + concat, the exchange unit's fused
# sum) instead of a cseg_amax_f32 pass per tensor; CSEG_PRODUCER_AMAX=0 restores the passes (A/B: profiles/r06_ab_producer_amax.txt)
PRODUCER_AMAX = os.environ.get("CSEG_PRODUCER_AMAX", "1") == "1"
# ... but NOT from the exchange unit's fused sum by default: its blocks are short (1 024 pixels) and there are 12 000 of them, and every
# publish is a request to one of the record's 32 cache lines -- measured inside the step 34.7 us per call with the record against 24.7 without,
# more than the 8 us pass it replaces. The entry point stays (cseg_fuse_sum_fwd_amax), tested, for callers with fat tiles.
FUSE_SUM_AMAX = os.environ.get("CSEG_FUSE_SUM_AMAX", "0") == "1"


class UpsampleConcat(Function):
    """lib/models/nets/hrnet.py:86-91 as one kernel (forward) and its exact adjoint (backward)."""

    @staticmethod
    def forward(ctx, amax, *feats):
        feats = [f.contiguous() for f in feats]
        B = feats[0].shape[0]
        C = [f.shape[1] for f in feats]
        hs = [f.shape[2] for f in feats]
        ws = [f.shape[3] for f in feats]
        out = torch.empty(B, sum(C), hs[0], ws[0], dtype=F32, device=feats[0].device)
        ptrs = (ctypes.c_void_p * len(feats))(*[_p(f, F32, "feat%d" % i).value for i, f in enumerate(feats)])
        if amax is not None:
            _hip.call("cseg_upcat_fwd_amax", ptrs, _int_arr(C), _int_arr(hs), _int_arr(ws), len(feats), B,
                      _p(out, F32, "out"), _pf(amax), _hip.stream_ptr())
        else:
            _hip.call("cseg_upcat_fwd", ptrs, _int_arr(C), _int_arr(hs), _int_arr(ws), len(feats), B,
                      _p(out, F32, "out"), _hip.stream_ptr())
        ctx.dims = (B, C, hs, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, hs, ws = ctx.dims
        g = g.contiguous()
        grads = []
        ptrs = []
        for i in range(len(C)):
            if ctx.needs_input_grad[i + 1]:
                t = torch.empty(B, C[i], hs[i], ws[i], dtype=F32, device=g.device)
                grads.append(t)
                ptrs.append(t.data_ptr())
            else:
                grads.append(None)
                ptrs.append(None)
        arr = (ctypes.c_void_p * len(C))(*ptrs)
        _hip.call("cseg_upcat_bwd", _p(g, F32, "d_out"), _int_arr(C), _int_arr(hs), _int_arr(ws), len(C), B, arr,
                  _hip.stream_ptr())
        return (None,) + tuple(grads)


def upsample_concat(feats):
    # the result feeds two split-operand convolutions (the head's 3x3 and the projection head's 1x1): its max|.| record is accumulated
    # by the kernel that writes it (a separate cseg_amax_f32 pass reads the 755 MB tensor once more: 0.15 ms)
    f0 = feats[0]
    amax = amax_request(f0) if (PRODUCER_AMAX and f0.shape[3] % 4 == 0 and f0.dtype == F32) else None
    return amax_attach(UpsampleConcat.apply(amax, *feats), amax)


class FuseSumReLU(Function):
    """out = relu(sum(same-resolution terms) + sum(bilinear-upsampled coarse terms)) in one kernel; the exchange
    step of HighResolutionModule.forward (hrnet_backbone.py:271-286 of the reference)."""

    @staticmethod
    def forward(ctx, n_same, amax, *terms):
        if n_same < 1:
            raise RuntimeError("fuse_sum_relu needs at least one same-resolution term")
        terms = [t.contiguous() for t in terms]
        same, low = terms[:n_same], terms[n_same:]
        B, C, h, w = same[0].shape
        out = torch.empty(B, C, h, w, dtype=F32, device=terms[0].device)
        sp = (ctypes.c_void_p * max(1, len(same)))(*[_p(t, F32, "same").value for t in same])
        lp = (ctypes.c_void_p * max(1, len(low)))(*([_p(t, F32, "low").value for t in low] or [None]))
        lh = [t.shape[2] for t in low]
        lw = [t.shape[3] for t in low]
        if amax is not None:
            _hip.call("cseg_fuse_sum_fwd_amax", sp, len(same), lp, _int_arr(lh or [1]), _int_arr(lw or [1]), len(low), B, C, h, w,
                      1, _p(out, F32, "out"), _pf(amax), _hip.stream_ptr())
        else:
            _hip.call("cseg_fuse_sum_fwd", sp, len(same), lp, _int_arr(lh or [1]), _int_arr(lw or [1]), len(low), B, C, h, w,
                      1, _p(out, F32, "out"), _hip.stream_ptr())
        ctx.save_for_backward(out)
        ctx.meta = (len(same), lh, lw, B, C, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        n_same, lh, lw, B, C, h, w = ctx.meta
        g = g.contiguous()
        need_same = any(ctx.needs_input_grad[2:2 + n_same])
        g_same = torch.empty_like(out) if need_same else None
        d_low = [torch.empty(B, C, lh[i], lw[i], dtype=F32, device=g.device)
                 if ctx.needs_input_grad[2 + n_same + i] else None for i in range(len(lh))]
        dl = (ctypes.c_void_p * max(1, len(d_low)))(*([t.data_ptr() if t is not None else None for t in d_low] or [None]))
        _hip.call("cseg_fuse_sum_bwd", _p(g, F32, "d_out"), _p(out, F32, "out"), _int_arr(lh or [1]), _int_arr(lw or [1]),
                  len(lh), B, C, h, w, _p(g_same, F32, "g_same") if need_same else _null(), dl, _hip.stream_ptr())
        grads = [g_same if ctx.needs_input_grad[2 + i] else None for i in range(n_same)]
        return (None, None) + tuple(grads) + tuple(d_low)


def affine_channels(u, a, b, want_amax=True):
    """a[c] + b[c] * u[b, c] over an NCHW tensor (one read + one write), leaving the max|.| record of the result for the split-operand
    convolution that reads it. -> (out, record or None)"""
    B, C = u.shape[:2]
    P = u[0, 0].numel()
    out = torch.empty_like(u)
    amax = amax_request(u) if want_amax else None
    _hip.call("cseg_affine_channels", _p(u, F32, "u"), _p(a.contiguous(), F32, "a"), _p(b.contiguous(), F32, "b"), B, C, ctypes.c_long(P),
              _pf(out), _pf(amax) if amax is not None else _null(), _hip.stream_ptr())
    return out, amax


def fuse_sum_relu(same, low):
    """same: list of [B,C,h,w]; low: list of [B,C,hs,ws] coarser maps (upsampled with align_corners=True)."""
    # the outputs of an exchange unit feed the split-operand convolutions of the next unit; the kernel CAN leave their max|.| record
    # (FUSE_SUM_AMAX), but by default kernels.amax_of spends a cseg_amax_f32 pass on each (~20 per step, 8 us): cheaper, see above
    amax = amax_request(same[0]) if FUSE_SUM_AMAX else None
    return amax_attach(FuseSumReLU.apply(len(same), amax, *same, *low), amax)


# ----------------------------------------------------------------------------------------------------------
# One gradient sum per branch of an exchange unit (round 4, opt-in: CSEG_FANOUT_SUM=1; not yet timed on the hardware).
# Every branch output of a HighResolutionModule feeds all of its (up to four) output resolutions (reference hrnet_backbone.py:271-286),
# so autograd accumulates up to four gradients per branch with `add` kernels: 72 launches and ~1.7 ms per step at batch 8
# (profiles/r04_step_steady_kernel_stats_epilogue_stats.csv), three read-read-write passes where one pass over the terms does.
# fan_out() hands every consumer its own alias of the tensor through ONE autograd node whose backward sums what arrives with the
# n-ary sum kernel of the exchange unit (cseg_fuse_sum_fwd without coarse terms and without the ReLU).
# ----------------------------------------------------------------------------------------------------------
# Round 6: ON by default. With the residual blocks on the grouped launches the step is one long chain on the compute stream and every
# launch in it is paid in wall time (before, the forked branches hid them): A/B/A/B on one MI355X 91.7 / 90.4 -> 86.3 / 85.4 ms per
# step (profiles/r06_ab_fanout_wgrad.txt; round 5, forked branches: 87.6 vs 87.5).
FANOUT_SUM = os.environ.get("CSEG_FANOUT_SUM", "1") == "1"


def sum_same(terms):
    """terms: 2..4 tensors [B,C,h,w] of one shape -> their sum, one kernel (no autograd)."""
    terms = [t.contiguous() for t in terms]
    B, C, h, w = terms[0].shape
    out = torch.empty(B, C, h, w, dtype=F32, device=terms[0].device)
    sp = (ctypes.c_void_p * len(terms))(*[_p(t, F32, "term").value for t in terms])
    lp = (ctypes.c_void_p * 1)(None)
    _hip.call("cseg_fuse_sum_fwd", sp, len(terms), lp, _int_arr([1]), _int_arr([1]), 0, B, C, h, w, 0, _p(out, F32, "out"),
              _hip.stream_ptr())
    return out


class FanOutSum(Function):
    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if len(gs) > 4 or gs[0].dim() != 4 or gs[0].dtype != F32 or not _on_device(gs[0]) or any(g.shape != gs[0].shape for g in gs):
            total = gs[0]
            for g in gs[1:]:
                total = total + g
            return total, None
        return sum_same(gs), None


def fan_out(x, n):
    """-> n aliases of x for n consumers, behind one autograd node that sums their gradients in one kernel. The max|.| record of x
    travels with every alias."""
    if n <= 1 or not x.requires_grad:
        return [x] * n
    outs = FanOutSum.apply(x, n)
    a = getattr(x, "_cseg_amax", None)
    if a is not None and a[1] == x._version:
        for o in outs:
            o._cseg_amax = (a[0], o._version) + tuple(a[2:])
    return list(outs)


# ----------------------------------------------------------------------------------------------------------
# segmentation term: upsample + weighted CE
# ----------------------------------------------------------------------------------------------------------
class UpsampleCE(Function):
    """FSCELoss(F.interpolate(seg, target.shape, bilinear, align_corners=True), target) without the [B,K,H,W]
    tensor (lib/loss/loss_contrast.py:180-181, lib/loss/loss_helper.py:169-206)."""

    @staticmethod
    def forward(ctx, seg, target, weight, ignore_index, status):
        seg = seg.contiguous()
        B, K, h, w = seg.shape
        _, H, W = target.shape
        lib = _hip.lib()
        nb = lib.cseg_upsample_ce_blocks(B, H, W)
        dev = seg.device
        partial = torch.empty(2 * nb, dtype=F32, device=dev)
        out = torch.empty(2, dtype=F32, device=dev)
        if status is None:
            status = torch.zeros(4, dtype=I32, device=dev)
        wp = _p(weight, F32, "ce_weight") if weight is not None else _null()
        lse = torch.empty(B, H, W, dtype=F32, device=dev)      # per-label-pixel log-sum-exp, reused by the backward
        _hip.call("cseg_upsample_ce_fwd", _p(seg, F32, "seg"), _p(target, I64, "target"), wp, int(ignore_index), B, K,
                  h, w, H, W, _p(partial, F32, "partial"), _p(out, F32, "out"), _p(status, I32, "status"),
                  _p(lse, F32, "lse"), _hip.stream_ptr())
        ctx.save_for_backward(seg, target, out, lse)
        ctx.weight = weight
        ctx.ignore_index = int(ignore_index)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        seg, target, out, lse = ctx.saved_tensors
        B, K, h, w = seg.shape
        _, H, W = target.shape
        d_seg = torch.empty_like(seg)
        g = g.reshape(1).to(F32).contiguous()
        wp = _p(ctx.weight, F32, "ce_weight") if ctx.weight is not None else _null()
        _hip.call("cseg_upsample_ce_bwd", _p(seg, F32, "seg"), _p(target, I64, "target"), wp, ctx.ignore_index, B, K,
                  h, w, H, W, _p(out, F32, "out"), _p(g, F32, "d_loss"), _p(lse, F32, "lse"), _p(d_seg, F32, "d_seg"),
              _hip.stream_ptr())
        return d_seg, None, None, None, None


def upsample_ce(seg, target, weight=None, ignore_index=-1, status=None):
    """`status` (i32 [4], optional, accumulated across calls): status[1] counts label values that are neither
    `ignore_index` nor in [0, K) -- nn.CrossEntropyLoss would assert on those; here they are dropped from numerator
    and denominator, and the caller is expected to look at the counter (FSCELoss.bad_label_count)."""
    return UpsampleCE.apply(seg, target, weight, ignore_index, status)


# ----------------------------------------------------------------------------------------------------------
# memory bank
# ----------------------------------------------------------------------------------------------------------
@torch.no_grad()
def queue_count(labels, stride, num_classes):
    B, H, W = labels.shape
    counts = torch.empty(B, num_classes, dtype=I32, device=labels.device)
    _hip.call("cseg_queue_count", _p(labels, I64, "labels"), B, H, W, int(stride), int(num_classes),
              _p(counts, I32, "counts"), _hip.stream_ptr())
    return counts


@torch.no_grad()
def queue_class_sums(keys, labels, stride, num_classes):
    B, D = keys.shape[:2]
    Pk = keys.shape[2] * keys.shape[3]
    _, H, W = labels.shape
    sums = torch.empty(B, num_classes, D, dtype=F32, device=keys.device)
    _hip.call("cseg_queue_class_sums", _p(keys, F32, "keys"), _p(labels, I64, "labels"), B, D, Pk, H, W, int(stride),
              int(num_classes), _p(sums, F32, "sums"), _hip.stream_ptr())
    return sums


@torch.no_grad()
def queue_write_segments(sums, counts, job_img, job_cls, job_dst_row, segment_queue):
    K, ms, D = segment_queue.shape
    _hip.call("cseg_queue_write_segments", _p(sums, F32, "sums"), _p(counts, I32, "counts"),
              _p(job_img, I32, "job_img"), _p(job_cls, I32, "job_cls"), _p(job_dst_row, I32, "job_dst_row"),
              job_img.numel(), K, D, _p(segment_queue, F32, "segment_queue"), ms, _hip.stream_ptr())
    torch.autograd.graph.increment_version(segment_queue)      # written through a raw pointer


@torch.no_grad()
def queue_write_pixels(keys, src_img, src_pos, dst_cls, dst_row, pixel_queue):
    B, D = keys.shape[:2]
    Pk = keys.shape[2] * keys.shape[3]
    K, ms, _ = pixel_queue.shape
    _hip.call("cseg_queue_write_pixels", _p(keys, F32, "keys"), B, D, Pk, _p(src_img, I32, "src_img"),
              _p(src_pos, I32, "src_pos"), _p(dst_cls, I32, "dst_cls"), _p(dst_row, I32, "dst_row"),
              src_img.numel(), _p(pixel_queue, F32, "pixel_queue"), ms, _hip.stream_ptr())
    torch.autograd.graph.increment_version(pixel_queue)


# ----------------------------------------------------------------------------------------------------------
# fused (Sync)BatchNorm + residual + ReLU primitives (host logic: lib/models/tools/fused_bn.py)
# ----------------------------------------------------------------------------------------------------------
F64 = torch.float64


def _opt(t, dtype, what):
    return _p(t, dtype, what) if t is not None else _null()


def _bn_dims(x):
    B, C = x.shape[0], x.shape[1]
    HW = 1
    for v in x.shape[2:]:
        HW *= v
    return B, C, HW


_BN_WS = {}      # (device index, stream) -> scratch, grown on demand
_BN_WS_NEED = {}  # (B, C, HW) -> floats of scratch the library wants


def _bn_ws(B, C, HW, device):
    """Reduction scratch of the BN kernels. Every kernel that writes it is followed on the SAME stream by the kernel that
    reads it, so one buffer per (device, stream) serves all layers (saves two allocator round trips per BN call: the
    step issues ~600 of them and is host-bound at small per-GPU batches)."""
    need = _BN_WS_NEED.get((B, C, HW))
    if need is None:                          # (a pure function of the three sizes: asked once per shape, not once per BN call)
        need = _BN_WS_NEED[(B, C, HW)] = max(1, _hip.lib().cseg_bn_ws_floats(B, C, HW))
    key = (device.index, _hip.raw_stream()) if device.type == "cuda" else (-1, 0)     # callers run on the current device (_hip.dev)
    buf = _BN_WS.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 16), dtype=F32, device=device)
        _BN_WS[key] = buf
    return buf


def bn_stats(x):
    """-> moments [C+1,2] f64: rows 0..C-1 = (sum x, sum x^2) over this rank's values, row C = (this rank's element count per
    channel, 0) -- the tensor a SyncBN exchange all-reduces; the summed row C is the global count (bn_finalize with count 0)."""
    B, C, HW = _bn_dims(x)
    moments = torch.empty(C + 1, 2, dtype=F64, device=x.device)
    _hip.call("cseg_bn_stats", _p(x, F32, "x"), B, C, HW, _p(_bn_ws(B, C, HW, x.device), F32, "ws"),
              _p(moments, F64, "moments"), _hip.stream_ptr())
    return moments


def bn_finalize(moments, count, eps, momentum, running_mean, running_var, num_batches_tracked):
    """(global) moments [C+1,2] -> mean_invstd [C,2] f32; running statistics / batch counter updated in place. count 0 = read the
    exchanged count from row C on the device (no host round trip, unequal per-rank batches allowed)."""
    C = moments.shape[0] - 1
    mi = torch.empty(C, 2, dtype=F32, device=moments.device)
    _hip.call("cseg_bn_finalize", _p(moments, F64, "moments"), C, float(count), float(eps), float(momentum),
              _opt(running_mean, F32, "running_mean"), _opt(running_var, F32, "running_var"),
              _opt(num_batches_tracked, I64, "num_batches_tracked"), _p(mi, F32, "mean_invstd"), _hip.stream_ptr())
    return mi


def bn_stats_finalize(x, eps, momentum, running_mean, running_var, num_batches_tracked):
    """Single-rank bn_stats + bn_finalize in two launches."""
    B, C, HW = _bn_dims(x)
    mi = torch.empty(C, 2, dtype=F32, device=x.device)
    _hip.call("cseg_bn_stats_finalize", _p(x, F32, "x"), B, C, HW, _p(_bn_ws(B, C, HW, x.device), F32, "ws"),
              float(eps), float(momentum), _opt(running_mean, F32, "running_mean"),
              _opt(running_var, F32, "running_var"), _opt(num_batches_tracked, I64, "num_batches_tracked"),
              _p(mi, F32, "mean_invstd"), _hip.stream_ptr())
    return mi


def bn_apply(x, mean_invstd, weight, bias, residual, relu, amax=None):
    """amax: zeroed word (amax_request) that receives max|y|."""
    B, C, HW = _bn_dims(x)
    y = torch.empty_like(x)
    _hip.call("cseg_bn_apply_amax", _pq(x, "x"), _pq(residual, "residual") if residual is not None else _null(), _pf(mean_invstd),
              _opt(weight, F32, "weight"), _opt(bias, F32, "bias"), int(bool(relu)), B, C, HW, _p(y, F32, "y"),
              _pf(amax) if amax is not None else _null(), _hip.stream_ptr())
    return y


def bn_bwd_reduce(dy, x, out, mean_invstd, weight, bias, mode):
    """-> (sums [C+1,2] f64 (row C = this rank's element count, 0), d_weight [C], d_bias [C], g_masked or None).
    mode: 0 none | 1 ReLU mask from x | 2 from out."""
    B, C, HW = _bn_dims(x)
    dev = x.device
    sums = torch.empty(C + 1, 2, dtype=F64, device=dev)
    d_weight = torch.empty(C, dtype=F32, device=dev)
    d_bias = torch.empty(C, dtype=F32, device=dev)
    g = torch.empty_like(x) if mode == 2 else None
    _hip.call("cseg_bn_bwd_reduce", _p(dy, F32, "dy"), _p(x, F32, "x"), _opt(out, F32, "out"),
              _p(mean_invstd, F32, "mean_invstd"), _opt(weight, F32, "weight"), _opt(bias, F32, "bias"), int(mode),
              B, C, HW, _p(_bn_ws(B, C, HW, dev), F32, "ws"), _opt(g, F32, "g_masked"), _p(sums, F64, "sums"),
              _p(d_weight, F32, "d_weight"), _p(d_bias, F32, "d_bias"), _hip.stream_ptr())
    return sums, d_weight, d_bias, g


def bn_bwd_apply(dy, x, mean_invstd, weight, bias, sums, count, mask_from_x, amax=None):
    """sums None = frozen statistics (eval mode); count 0 = the exchanged count in row C of `sums`. amax: zeroed word that
    receives max|dx|."""
    B, C, HW = _bn_dims(x)
    dx = torch.empty_like(x)
    _hip.call("cseg_bn_bwd_apply_amax", _p(dy, F32, "dy"), _p(x, F32, "x"), _p(mean_invstd, F32, "mean_invstd"),
              _opt(weight, F32, "weight"), _opt(bias, F32, "bias"), _opt(sums, F64, "sums"), float(count),
              int(bool(mask_from_x)), B, C, HW, _p(dx, F32, "dx"), _pf(amax) if amax is not None else _null(),
              _hip.stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------------------------
# 3x3 / stride 1 / pad 1 convolution of the narrow HRNet branches (csrc/conv3x3.hip)
# ----------------------------------------------------------------------------------------------------------
def conv3x3_eligible(x, weight):
    """Shapes the MFMA kernel covers: NCHW fp32 on the GPU, 3x3, input channels % 8, output channels % 48 (both ways:
    backward-data swaps them), width % 4."""
    if not (_on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4 and x.is_contiguous()):
        return False
    co, ci, kh, kw = weight.shape
    return (kh, kw) == (3, 3) and ci % 48 == 0 and co % 48 == 0 and x.shape[1] == ci and x.shape[3] % 4 == 0


@torch.no_grad()
def _conv3x3_run(x, weight, transpose_flip):
    co, ci = weight.shape[:2]
    conv_in, conv_out = (co, ci) if transpose_flip else (ci, co)
    B, _, H, W = x.shape
    lib = _hip.lib()
    wp = torch.empty(lib.cseg_conv3x3_packed_floats(conv_in, conv_out), dtype=F32, device=x.device)
    sp = _hip.stream_ptr()
    _hip.call("cseg_conv3x3_pack_weights", _p(weight, F32, "weight"), co, ci, int(transpose_flip), _pf(wp), sp)
    y = torch.empty(B, conv_out, H, W, dtype=F32, device=x.device)
    _hip.call("cseg_conv3x3_fwd", _p(x, F32, "x"), _pf(wp), B, conv_in, conv_out, H, W, _pf(y), sp)
    return y


@torch.no_grad()
def _conv3x3_wrw(x, dy, co, ci):
    B, _, H, W = x.shape
    lib = _hip.lib()
    ws = torch.empty(lib.cseg_conv3x3_wrw_ws_floats(B, ci, co, H, W), dtype=F32, device=x.device)
    dw = torch.empty(co, ci, 3, 3, dtype=F32, device=x.device)
    _hip.call("cseg_conv3x3_wrw", _p(x, F32, "x"), _p(dy, F32, "dy"), B, ci, co, H, W, _pf(ws), _pf(dw),
              _hip.stream_ptr())
    return dw


# Weight gradient on the MFMA kernel only where it beats MIOpen's NHWC implicit-GEMM + 3 layout transposes
# (tools/conv3x3_probe.py on MI355X, bs 8: 48 ch 141 vs 182 us; 96 ch 133 vs 139 us; 192 ch 132 vs 123 us)
CONV3X3_WRW_CHANNELS = (48, 96)
# Forward on the MFMA kernel only when there are enough 4x64 output tiles to fill the chip (bs 8: 1024 tiles, 114 vs
# 168 us; one image: 128 tiles on 256 CUs, 35 vs 25 us -> MIOpen). Backward-data and the weight gradient win at both.
CONV3X3_MIN_FWD_TILES = 256


class Conv3x3(Function):
    """y = conv2d(x, weight, stride 1, padding 1): forward, backward-data and the weight gradient on the MFMA kernels."""

    @staticmethod
    def forward(ctx, x, weight):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        B, _, H, W = x.shape
        n_tiles = B * (weight.shape[0] // 48) * ((H + 3) // 4) * ((W + 63) // 64)
        if n_tiles < CONV3X3_MIN_FWD_TILES:
            return torch.nn.functional.conv2d(x, weight, None, 1, 1)          # MIOpen (same math, fp32)
        return _conv3x3_run(x, weight, False)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _conv3x3_run(dy, weight, True) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if weight.shape[0] == weight.shape[1] and weight.shape[0] in CONV3X3_WRW_CHANNELS:
                dw = _conv3x3_wrw(x, dy, weight.shape[0], weight.shape[1])
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return dx, dw


def conv3x3(x, weight):
    return Conv3x3.apply(x, weight)


# ----------------------------------------------------------------------------------------------------------
# Arithmetic of the split-operand convolutions (csrc/cseg_split.h): "f16x3" (default since round 3: two scaled fp16 pieces per
# operand, three MFMAs per product, roof 2500/3 = 833 TFLOP/s of fp32-equivalent work) or "bf16x6" (round 2: three bf16 pieces,
# six MFMAs, 417 TFLOP/s). Both are fp32-class: whole-network logits vs fp64 5.1e-5 / 2.1e-5 against fp32's own 4.3e-5
# (tools/split_bf16_probe.py). f16x3 needs max|tensor| of every operand: a device-side uint32 (bit pattern of the float)
# that cseg_amax_f32 accumulates and the kernels read -- no host round trip.
# ----------------------------------------------------------------------------------------------------------
ARITH_IDS = {"bf16x6": 0, "f16x3": 1}
SPLIT_ARITH = os.environ.get("CSEG_SPLIT_ARITH", "f16x3")
if SPLIT_ARITH not in ARITH_IDS:
    raise RuntimeError("CSEG_SPLIT_ARITH must be one of %s (got %r)" % (sorted(ARITH_IDS), SPLIT_ARITH))


def split_arith_id():
    return ARITH_IDS[SPLIT_ARITH]


AMAX_WORDS = 1024        # CSEG_AMAX_WORDS of include/cseg_hip.h: 32 slots, 128 bytes apart
AMAX_ARENA_RECORDS = 2048
_AMAX_ARENAS = {}        # device -> [records: AMAX_ARENA_RECORDS views of one int32 arena [.. x AMAX_WORDS] (zeroed once), next free record, zero-fill event, streams that waited for it]


def amax_slot(device):
    """A zeroed max|.| record on the device (include/cseg_hip.h: CSEG_AMAX_WORDS uint32): a view into an arena that is zero-filled
    once per 2048 records, so that a maximum costs ONE launch, not a fill + a launch. Records are not recycled: a used arena
    lives as long as a view of it.
    Streams: the zero fill runs on whatever stream is current when an arena is started, and the records are handed to kernels on
    ANY stream (the forked HRNet branches, the mining side stream, the autograd engine's replay of those forks). Every stream
    therefore waits ONCE per arena for the fill's event before it takes its first record from it (round 4: with the branches on
    four streams an arena switch in the middle of a fork let three of them accumulate into records that were zeroed afterwards --
    the step's loss moved in the second digit, run to run)."""
    key = (device.type, device.index)
    st = _AMAX_ARENAS.get(key)
    if st is None or st[1] >= len(st[0]):
        # the records as a tuple of views made in ONE call: indexing the arena tensor per record (`arena[i]`) costs 10-30 us of host
        # time each, ~1 000 records per step (tools/host_null_bench.py: amax_request 32 -> 1 us)
        arena = torch.zeros(AMAX_ARENA_RECORDS, AMAX_WORDS, dtype=I32, device=device).unbind(0)
        ev = None
        # (not while a hipGraph is being captured: the capture starts its arena on the capturing stream before any fork, and every
        # forked stream joins the capture by waiting for that stream -- the fill is ordered before them by the graph's own edges)
        if device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
        st = [arena, 0, ev, {_hip.raw_stream()} if ev is not None else None]
        _AMAX_ARENAS[key] = st
    if st[2] is not None:
        sid = _hip.raw_stream()
        if sid not in st[3]:
            torch.cuda.current_stream(device).wait_event(st[2])
            st[3].add(sid)
    i = st[1]
    st[1] = i + 1
    return st[0][i]


def tensor_amax(t, slot=None):
    """max|t| as the split kernels take it: a device record (amax_slot) whose maximum word is the bit pattern of that float.
    `slot`: accumulate into an existing record (max over several tensors)."""
    if slot is None:
        slot = amax_slot(t.device)
    _hip.call("cseg_amax_f32", _p(t, F32, "tensor"), ctypes.c_long(t.numel()), _pf(slot), _hip.stream_ptr())
    return slot


def amax_request(t):
    """A zeroed max|.| word for a tensor a producer kernel is about to write (the fused-BN apply kernels accumulate it while
    they store), or None when nobody will read it (bf16x6 arithmetic)."""
    return amax_slot(t.device) if (split_arith_id() and _on_device(t)) else None


def amax_attach(t, slot):
    """Hands the word to whoever consumes `t` next: an attribute of the tensor object (it travels with the tensor through
    autograd; the version counter guards against a later in-place change of the values)."""
    if slot is not None:
        t._cseg_amax = (slot, t._version)
    return t


def known_amax(t):
    """The max|t| record that travels with `t`, or None. A record that a CONSUMER computed (amax_of below) was produced on that
    consumer's stream, after `t` itself: another stream that picks it up first waits for the pass (round 4: with the two heads of
    the segmentor on different streams the classifier head read the record of their common input before the projection head's
    pass over it had run -- a scale from an empty record, NaN losses)."""
    a = getattr(t, "_cseg_amax", None)
    if a is None or a[1] != t._version:
        return None
    if len(a) > 2 and a[3] is not None and a[2] != _hip.raw_stream():
        torch.cuda.current_stream(t.device).wait_event(a[3])
    return a[0]


def amax_of(t):
    """max|t| word: the producer's if it left one, a pass over the tensor otherwise."""
    a = known_amax(t)
    if a is None:
        a = tensor_amax(t.contiguous())
        try:
            # a second consumer of the same tensor (the head's 3x3 and 1x1 both read `feats`) reuses it -- after waiting for this pass
            # when it runs on another stream
            if t.is_cuda and not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(t.device))
                t._cseg_amax = (a, t._version, _hip.raw_stream(), ev)
            else:
                amax_attach(t, a)
        except Exception:
            pass
    return a


# ----------------------------------------------------------------------------------------------------------
# BatchNorm statistics from the producing convolution's epilogue (csrc/cseg_stats.h, round 4): the `_st` forward entry points write
# per-(channel, 64-pixel segment) (count, mean, M2) records while the output is still in registers; the record buffer travels with
# the output tensor exactly like the max|.| word above, and fused_bn.bn_forward finalises it (cseg_bn_tiles_finalize /
# cseg_bn_tiles_moments) instead of reading the whole tensor once more (cseg_bn_stats*: 306 launches, 2.6 ms per step in round 3).
# ----------------------------------------------------------------------------------------------------------
CONV_EPILOGUE_STATS = os.environ.get("CSEG_CONV_STATS", "1") == "1"


_STAT_SEGMENTS = {}


def tile_stats_buffer(kind, c_out, B, H, W, device):
    """[Cout, T, 4] f32 for the statistics epilogue of a forward convolution whose OUTPUT is [B, Cout, H, W] (kind 0: 3x3 kernels,
    1: 1x1 kernels)."""
    key = (kind, B, H, W)
    T = _STAT_SEGMENTS.get(key)
    if T is None:                             # (a pure function of the output's shape)
        T = _STAT_SEGMENTS[key] = _hip.lib().cseg_conv_stat_segments(int(kind), int(B), int(H), int(W))
    return torch.empty(c_out, T, 4, dtype=F32, device=device)


def tile_stats_attach(t, stats):
    if stats is not None:
        t._cseg_tile_stats = (stats, t._version)
    return t


def known_tile_stats(t):
    a = getattr(t, "_cseg_tile_stats", None)
    return a[0] if (a is not None and a[1] == t._version) else None


def bn_tiles_finalize(stats, eps, momentum, running_mean, running_var, num_batches_tracked):
    """Epilogue statistics [C, T, 4] -> mean_invstd [C,2]; running statistics / batch counter updated like bn_stats_finalize."""
    C, T = stats.shape[0], stats.shape[1]
    mi = torch.empty(C, 2, dtype=F32, device=stats.device)
    _hip.call("cseg_bn_tiles_finalize", _pf(stats), C, ctypes.c_long(T), float(eps), float(momentum),
              _opt(running_mean, F32, "running_mean"), _opt(running_var, F32, "running_var"),
              _opt(num_batches_tracked, I64, "num_batches_tracked"), _pf(mi), _hip.stream_ptr())
    return mi


def bn_tiles_moments(stats):
    """Epilogue statistics [C, T, 4] -> moments [C+1,2] f64 (what bn_stats returns: the tensor a SyncBN exchange all-reduces)."""
    C, T = stats.shape[0], stats.shape[1]
    moments = torch.empty(C + 1, 2, dtype=torch.float64, device=stats.device)
    _hip.call("cseg_bn_tiles_moments", _pf(stats), C, ctypes.c_long(T), _pf(moments), _hip.stream_ptr())
    return moments


# ----------------------------------------------------------------------------------------------------------
# Weight gradients off the critical path (round 4). In backward, dx of a convolution feeds the next node of the chain; dw feeds
# nothing until the optimizer runs. With `wgrad_scope` active (Trainer.train_step opens it around loss.backward() on single-rank
# GPU runs) the split-operand weight-gradient kernels go to ONE side stream: it waits for the stream of the backward node (dy and
# the max|.| words are ready there), and the trainer joins it before the optimizer step. MFMA-bound weight gradients then share the
# chip with the HBM-bound BatchNorm passes and small backward-data launches of the chain instead of queueing between them.
# Off outside the scope (a caller that reads .grad right after backward() would not know about the side stream), while a hipGraph
# is being captured, and under DDP (its reducer copies gradients on the backward stream as they appear).
# MEASURED (GPU call r04j11, profiles/r04_branch_streams_ab.txt): with the branches already forked this stream makes the step SLOWER
# (91.1 / 89.6 ms with it, 88.7 / 86.6 without: the weight gradients then compete with the chain for the same CUs instead of
# filling holes) -- so it is an opt-in experiment (CSEG_WGRAD_STREAM=1), gradients verified equal by tests/test_gpu_streams.py.
# ----------------------------------------------------------------------------------------------------------
# Round 6: measured again on the serial chain of the grouped launches (the grouped weight gradients go through _group_wrw): the
# weight gradients (matrix-bound) then run beside the BatchNorm passes (HBM-bound) of the next block -- 89.0 / 92.0 -> 87.9 / 87.8 ms per
# step and, second box, 86.6 / 88.2 -> 85.5 / 86.4 (A/B/A/B, profiles/r06_ab_fanout_wgrad.txt): about -1.2 ms. STILL opt-in: a process
# that ran eager steps with this stream and then captures a step graph (tests/test_gpu_step_graph.py runs both in one process) got
# ZERO parameter gradients from the replay (71 and 920 of 920 tensors in two runs: a race, GPU calls r06_g16 / r06_g18; with the switch
# off the same test passes) -- not understood, so not a default.
WGRAD_STREAM = os.environ.get("CSEG_WGRAD_STREAM", "0") == "1"
_WGRAD = {"on": False, "stream": None, "main": None, "used": False}


class wgrad_scope(object):
    """with kernels.wgrad_scope(device): loss.backward()   -- joins the weight-gradient stream on exit."""

    def __init__(self, device):
        self.active = bool(WGRAD_STREAM and device is not None and device.type == "cuda"
                           and not (torch.distributed.is_available() and torch.distributed.is_initialized()))
        self.device = device

    def __enter__(self):
        if self.active:
            if _WGRAD["stream"] is None or _WGRAD["stream"].device != self.device:
                _WGRAD["stream"] = torch.cuda.Stream(device=self.device)
            _WGRAD.update(on=True, main=torch.cuda.current_stream(self.device), used=False)
        return self

    def __exit__(self, *exc):
        if self.active:
            _WGRAD["on"] = False
            if _WGRAD["used"]:
                _WGRAD["main"].wait_stream(_WGRAD["stream"])
        return False


def _on_wgrad_stream(fn, *inputs):
    """fn() -> one tensor (a weight gradient). Runs it on the weight-gradient stream when the scope is open; `inputs` are the
    tensors it reads (kept from being recycled until that stream is done with them)."""
    if not _WGRAD["on"] or not inputs[0].is_cuda or torch.cuda.is_current_stream_capturing():
        return fn()
    cur = torch.cuda.current_stream(inputs[0].device)
    side = _WGRAD["stream"]
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    out.record_stream(_WGRAD["main"])        # read by the optimizer on the trainer's stream, after wgrad_scope joined the side stream
    _WGRAD["used"] = True
    return out


import weakref

import numpy as np

_JOB_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("amax", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("flag", "<i4"),
                       ("nt", "<i4"), ("kind", "<i4"), ("total", "<i4"), ("block0", "<i4"), ("reserved", "<i4")])   # cseg_split_job


class SplitWeights(object):
    """Packed forms of every split-operand convolution weight, refreshed for the WHOLE network in three launches per optimizer
    step (zero the max|w| records, cseg_amax_batch, cseg_split_pack_batch) instead of three launches per layer (round-3 trace:
    213 + 426 launches of ~5 us = 3.9 ms of GPU time and as many host launches per step).
    A pack request (weight, operator) is registered on first use; its buffer and the weight's max|w| record then live as long as
    the weight. Staleness = the weight's (storage pointer, version counter, epoch) differs from what was packed. The version
    counter alone is NOT enough: torch's fused optimizers (torch._fused_sgd_ & co.) and every write through `param.data` or a raw
    pointer change the values without bumping `Tensor._version` (with fused SGD the step-0 packs stayed in use for the
    whole run). So there is an explicit `epoch`, advanced by `invalidate()`:
      * after EVERY optimizer step of ANY torch optimizer (a global step post-hook registered below),
      * by ModuleRunner.load_state_dict (nn.Module.load_state_dict itself copies with `copy_`, which does bump the version),
      * by the caller after any other out-of-band write (EMA / SWA copies through `.data`, manual re-initialisation).
    The first request after that repacks every entry at once (three launches for the whole network)."""

    def __init__(self):
        self.weights = {}                                # id(weight) -> state dict (holds a weak reference; dropped with the weight)
        self.arenas = {}                                 # device -> [arena [n, AMAX_WORDS] int32 (one record per weight), next free row]
        self.table_cache = {}                            # (device, kind of table) -> (identity tuple, device tensor, n_jobs, total_blocks)
        self.epoch = 0
        self.generation = 0                              # bumped when record ADDRESSES change (_grow): captured hipGraphs hold them
        self.retired = []                                # outgrown arenas stay allocated: a graph captured earlier may still read them

    def invalidate(self):
        """Marks every packed weight (and its max|w| record) stale: call after any change of weight VALUES that does not go through
        an autograd-visible in-place op. Costs nothing until the next request."""
        self.epoch += 1

    @staticmethod
    def _dkey(device):
        return (device.type, device.index)

    def _record(self, state):
        """The max|w| record of a registered weight: a view made once per (weight, arena) -- `arena[row]` per call was 5-10 us of the
        10 us a get() cost on the host, four get() per residual block and step."""
        arena = self.arenas[state["dev"]][0]
        rec = state.get("rec")
        if rec is None or state.get("rec_of") is not arena:
            rec = arena[state["row"]]
            state["rec"], state["rec_of"] = rec, arena
        return rec

    def _grow(self, device):
        """A larger record arena for `device` (one arena per device: two GPUs in one process keep separate tables)."""
        key = self._dkey(device)
        old = self.arenas.get(key)
        n = 0 if old is None else old[0].shape[0]
        # 2 048 records (8 MB) to start with: HRNet-W48-contrast registers ~330 weights, a second model in the same process still fits
        new = torch.zeros(max(2048, 2 * n), AMAX_WORDS, dtype=I32, device=device)
        if n:
            new[:n].copy_(old[0])
            # a hipGraph captured before this point has the OLD record addresses baked in. The old arena is kept alive (no
            # read of freed memory) and `generation` tells segmentor/tools/step_graph.py to capture again (its records are refreshed
            # in the new arena only).
            self.retired.append(old[0])
            self.generation += 1
        self.arenas[key] = [new, 0 if old is None else old[1]]
        for st in self.weights.values():
            if st["dev"] == key:
                st["amax_of"] = None if n == 0 else st.get("amax_of")      # (the old records were copied into the new arena)
                for e in st["entries"].values():
                    e["version"] = None                  # record addresses changed: every entry of this device is packed again
        for k in [k for k in self.table_cache if k[0] == key]:
            del self.table_cache[k]

    def get(self, weight, tag, flag, nt_req):
        """-> (packed buffer uint8, max|w| record or None). tag: 'c3' | 'c1'."""
        arith = split_arith_id()
        st = self.weights.get(id(weight))
        dkey = self._dkey(weight.device)
        if st is None or st["ref"]() is not weight or st["dev"] != dkey:
            ar = self.arenas.get(dkey)
            if ar is None or ar[1] >= ar[0].shape[0]:
                self._grow(weight.device)
                ar = self.arenas[dkey]
            wid = id(weight)
            st = {"row": ar[1], "dev": dkey, "entries": {},
                  "ref": weakref.ref(weight, lambda _r, wid=wid: self.weights.pop(wid, None))}
            ar[1] += 1                                   # rows are not recycled (a dead weight's row stays zero)
            self.weights[wid] = st
        key = (tag, bool(flag), int(nt_req), SPLIT_ARITH, os.environ.get("CSEG_CONV3X3_SB_VAR"), os.environ.get("CSEG_CONV3X3_SB16_CH"))
        e = st["entries"].get(key)
        if e is None:
            e = self._plan(weight, tag, flag, nt_req, arith)
            st["entries"][key] = e
        now = (weight.data_ptr(), weight._version, self.epoch)
        if e["version"] != now:
            self.refresh(weight.device)
            if e["version"] != now:
                raise RuntimeError("SplitWeights: the packed form of a %s weight on %s is still stale after a refresh"
                                   % (tuple(weight.shape), weight.device))
        return e["wp"], (self._record(st) if arith else None)

    def _plan(self, weight, tag, flag, nt_req, arith):
        co, ci = weight.shape[:2]
        conv_in, conv_out = (co, ci) if flag else (ci, co)
        lib = _hip.lib()
        kind, nt, threads = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_long(0)
        if tag == "c3":
            ok = lib.cseg_conv3x3_split_plan(conv_in, conv_out, int(nt_req), ctypes.byref(kind), ctypes.byref(nt), ctypes.byref(threads))
            n_bytes = lib.cseg_conv3x3_split_packed_bytes(arith, conv_in, conv_out)
        elif tag == "c3s2":                              # stride 2: flag = backward-data operator, nt as requested
            ok = arith == ARITH_IDS["f16x3"] and lib.cseg_conv3x3_s2_split_plan(conv_in, conv_out, int(bool(flag)), int(nt_req),
                                                                                 ctypes.byref(kind), ctypes.byref(threads))
            nt = ctypes.c_int(int(nt_req))
            n_bytes = lib.cseg_conv3x3_s2_split_packed_bytes(conv_in, conv_out)
        else:
            ok = lib.cseg_conv1x1_split_plan_arith(arith, conv_in, conv_out, ctypes.byref(nt), ctypes.byref(threads))
            kind = ctypes.c_int(2)
            n_bytes = lib.cseg_conv1x1_split_packed_bytes(arith, conv_in, conv_out)
        if not ok or n_bytes == 0:
            raise RuntimeError("split convolution: unsupported channel counts %d -> %d (%s)" % (conv_in, conv_out, tag))
        wp = torch.empty(n_bytes, dtype=torch.uint8, device=weight.device)
        return {"wp": wp, "wp_ptr": wp.data_ptr(), "kind": kind.value, "nt": nt.value, "total": threads.value, "flag": int(bool(flag)),
                "version": None, "arith": arith}

    def refresh_all(self):
        """refresh() for every device that holds registered weights: what a hipGraph replay calls before its forward graph (the
        replay runs none of the Python that would otherwise notice stale packs on first use)."""
        for (kind, index), st in list(self.arenas.items()):
            self.refresh(st[0].device)

    @torch.no_grad()
    def refresh(self, device):
        """Host cost matters here: this runs at the top of every step, when the GPU has nothing queued behind the optimizer kernels
        (tools/host_profile.py: 2.7 ms per step for the first version, which rebuilt ~1 000 Python tuples and ~1 300 tensor views per
        step only to find the cached device tables unchanged). Now one pass over the entries collects what identifies the tables BY
        VALUE (pointers, record rows, formats); the row tuples are only built on a cache miss.
        Round 6: a max|w| record is recomputed only when ITS weight changed (state "amax_of" = the (pointer, version, epoch) the
        record was accumulated from). Before, ANY stale pack -- e.g. a packed form requested for the first time in the middle of the
        first step: the transposed operators of the backward pass, a new tiling -- zero-filled the WHOLE arena and re-accumulated every
        record on the requesting stream, while kernels of other streams (the forked exchange paths, autograd's concurrent replay of
        them) were reading their layers' records: a scale taken from a half-accumulated record. Seen as a first-step gradient that
        differed between the forked and the single-stream run (tests/test_gpu_streams.py) once the grouped launches moved the first
        requests of a step onto the main stream. At the top of a step every weight is stale and nothing else runs: one fill + one
        launch for all, as before."""
        arith = split_arith_id()
        dkey = self._dkey(device)
        arena = self.arenas[dkey][0]
        base = arena.data_ptr()
        epoch = self.epoch
        stale, every, need_amax = [], [], []
        for st in list(self.weights.values()):               # (a copy: weak-reference callbacks may drop entries meanwhile)
            w = st["ref"]()
            if w is None or st["dev"] != dkey or not _on_device(w):
                continue
            ptr = w.data_ptr()
            now = (ptr, w._version, epoch)
            every.append((ptr, w.numel(), st["row"]))
            if st.get("amax_of") != now:
                need_amax.append((ptr, w.numel(), st["row"], st, now))
            for e in st["entries"].values():
                if e["version"] != now and e["arith"] == arith:
                    stale.append((w, st, e, now))
        if not stale:
            return
        sp = _hip.stream_ptr()
        rec = lambda row: base + row * AMAX_WORDS * 4                    # device address of a max|w| record (int32 words)
        if arith and need_amax:
            if len(need_amax) == len(every):
                # every record is re-accumulated (140 MB of weights: ~30 us), so one fill serves them all
                arena.zero_()
                tab = self._table(dkey, "amax", tuple(every),
                                  lambda: [(ptr, 0, rec(row), 0, 0, 0, 0, 0, n, max(1, min(64, n // 16384))) for ptr, n, row in every])
            else:
                # some weights only (a layer used for the first time mid-step, an out-of-band write to a few tensors): their rows alone
                rows = torch.tensor([r[2] for r in need_amax], dtype=torch.int64, device=arena.device)
                arena.index_fill_(0, rows, 0)
                ident = tuple((ptr, n, row) for ptr, n, row, _, _ in need_amax)
                tab = self._table(dkey, "amax_some", ident,
                                  lambda: [(ptr, 0, rec(row), 0, 0, 0, 0, 0, n, max(1, min(64, n // 16384))) for ptr, n, row in ident])
            _hip.call("cseg_amax_batch", tab[0].data_ptr(), tab[1], tab[2], sp)
            for _, _, _, st, now in need_amax:
                st["amax_of"] = now
        tab = self._table(dkey, "pack", tuple((now[0], e["wp_ptr"], st["row"], e["flag"], e["nt"], e["kind"], e["total"]) for _, st, e, now in stale),
                          lambda: [(now[0], e["wp_ptr"], rec(st["row"]) if arith else 0, w.shape[0], w.shape[1], e["flag"], e["nt"],
                                    e["kind"], e["total"], (e["total"] + 255) // 256) for w, st, e, now in stale])
        _hip.call("cseg_split_pack_batch", tab[0].data_ptr(), tab[1], tab[2], arith, sp)
        for _, _, e, now in stale:
            e["version"] = now

    def _table(self, dkey, name, identity, make_rows):
        """make_rows() -> rows (src, dst, amax, cout, cin, flag, nt, kind, total, n_blocks); returns (device table, n_jobs,
        total_blocks). The device copy is reused while the same jobs come back (every step of a training run). `identity` must name
        everything the table holds by VALUE -- pointers and record rows, not Python object ids (those are recycled: a new layer that
        landed on a dead layer's id and storage would otherwise inherit its max|w| row, and a zero row means an overflowing scale)."""
        name = (dkey, name)
        hit = self.table_cache.get(name)
        if hit is not None and hit[0] == identity:
            return hit[1:]
        rows = make_rows()
        arr = np.zeros(len(rows), dtype=_JOB_DTYPE)
        b0 = 0
        for i, r in enumerate(rows):
            arr[i] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], b0, 0)
            b0 += r[9]
        dev = torch.from_numpy(arr.view(np.uint8).copy()).to(self.arenas[dkey][0].device)
        self.table_cache[name] = (identity, dev, len(rows), b0)
        return dev, len(rows), b0


SPLIT_WEIGHTS = SplitWeights()


def invalidate_packed_weights(*_args, **_kwargs):
    """Hook form of SPLIT_WEIGHTS.invalidate() (looked up at call time: tests replace the instance)."""
    SPLIT_WEIGHTS.invalidate()


# Every optimizer step marks the packs stale -- whatever the optimizer implementation does to Tensor._version (torch's fused
# kernels do not touch it).
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
_register_step_hook(invalidate_packed_weights)


# ----------------------------------------------------------------------------------------------------------
# The same convolution on the BF16 matrix cores with split operands (csrc/conv3x3_sb.hip); on by default
# ----------------------------------------------------------------------------------------------------------
# module-level switch (read at call time by the modules, so bench.py / tests can flip it inside one process)
CONV3X3_SPLIT_BF16 = os.environ.get("CSEG_CONV3X3_SPLIT_BF16", "1") == "1"
# bias-free residual-branch convolutions that move to the split kernel when the switch is on (measured on MI355X at the
# benched shapes, tools/conv3x3_sb_probe.py: 96 ch 70 vs 101 us; 48 ch 99 vs 106 us on the fp32-MFMA kernel; 192 ch at
# 8x32x64 with 3 channel tiles per block 81 / 78 us forward / backward-data vs 98-117 us on MIOpen's Winograd kernel,
# profiles/r02_conv3x3_split_bf16_nt_probe.jsonl). 384 channels (16x32 maps) can be added but waste half of every tile.
# 384 channels (8x16x32 maps: half of every 4x64 tile is padding) since round 3: with three MFMAs per product the kernel still beats
# MIOpen's NHWC implicit GEMM + its layout transposes (87 vs 117 + ~30 us, gpurun r03j3 split_arith_probe).
CONV3X3_SB_BRANCH_CHANNELS = tuple(int(c) for c in os.environ.get("CSEG_CONV3X3_SB_CHANNELS", "48,64,96,192,384").split(","))
# channel counts that go through the explicit-tiling entry points (conv3x3_sb_pick_nt); 48 / 96 / 720 keep the library's
# default tiling, which is what the parity suite ran on
CONV3X3_SB_PICK_NT_CHANNELS = (192, 384)


def conv3x3_sb_eligible(x, weight):
    """NCHW fp32 on the GPU, 3x3, channels % 48 (or exactly 64: the layer-1 bottlenecks) both ways (forward needs Cin % 16 and
    a tiling Cout, backward-data the mirror image)."""
    if not (_on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4 and x.is_contiguous()):
        return False
    co, ci, kh, kw = weight.shape
    ok = lambda c: c % 48 == 0 or c % 64 == 0
    # (any width since round 5: rows that are not 16-byte aligned are stored element by element, cseg_store_row4)
    return (kh, kw) == (3, 3) and ok(ci) and ok(co) and x.shape[1] == ci


# nt value that selects the 8 x 64-pixel kernel (include/cseg_hip.h: CSEG_NT_SB8) for wide layers: f16x3, output channels % 144
NT_SB8 = 0x109
CONV3X3_SB8 = os.environ.get("CSEG_CONV3X3_SB8", "1") == "1"


def conv3x3_sb_head_nt(c_out, x=None):
    """nt for the layers whose operator has `c_out` output channels: NT_SB8 where the 8-row kernel applies AND pays, else 0 (library
    default). Measured on the MI355X (tools/sb8_probe.py, profiles/r03_sb8_probe.jsonl): 720 -> 720 at 8 x 128 x 256 (2 560 blocks)
    5.28 vs 5.41 ms forward, 5.59 vs 5.74 backward-data; at batch 1 (320 blocks: 1.25 rounds of 256 CUs) 0.99 vs 0.88 ms and at
    144 channels on 64 x 128 maps 124 vs 76 us -- so only launches of at least four rounds take it."""
    if not (CONV3X3_SB8 and SPLIT_ARITH == "f16x3" and c_out % 144 == 0):
        return 0
    if x is not None and x.shape[0] * ((x.shape[2] + 7) // 8) * ((x.shape[3] + 63) // 64) * (c_out // 144) < 1024:
        return 0
    return NT_SB8


def conv3x3_sb_pick_nt(x, c_out):
    """16-channel tiles per block: the largest of 9 / 6 / 3 that still gives the grid >= 256 blocks (one per CU), else
    the smallest that divides the channel count. Measured at 192 channels, 8x32x64: nt 3 (256 blocks) 81 us, nt 6 (128
    blocks) 113 us; at 96 channels, 8x64x128: nt 6 (256 blocks) 70 us, nt 3 89 us."""
    cands = [nt for nt in (9, 6, 3) if c_out % (16 * nt) == 0]
    spatial = x.shape[0] * ((x.shape[2] + 3) // 4) * ((x.shape[3] + 63) // 64)
    for nt in cands:
        if spatial * (c_out // (16 * nt)) >= 256:
            return nt
    return cands[-1]


def conv3x3_sb_pack(weight, transpose_flip=False, nt=0):
    """-> (wp uint8 [...], aw): weight split + packed for the forward (or, transpose_flip, the backward-data) operator in the
    current arithmetic; aw = max|w| record (None with bf16x6). Kept fresh by SplitWeights: one batched launch per optimizer step
    for all layers."""
    return SPLIT_WEIGHTS.get(weight, "c3", transpose_flip, nt)


def conv3x3_sb_run(x, weight, transpose_flip=False, bias=None, nt=0, ax=None, addend=None, want_stats=False):
    """y = conv2d(x, weight, bias, 1, 1) (transpose_flip: the backward-data operator of that convolution applied to x)
    through the split-operand MFMA kernel. nt = 0: the library's default channel tiling; 3 / 6 / 9: explicit. ax: max|x| word
    (tensor_amax) when the caller already has it; computed here otherwise (f16x3 only). addend: tensor of the output's shape added
    in the kernel's epilogue."""
    co, ci = weight.shape[:2]
    conv_in, conv_out = (co, ci) if transpose_flip else (ci, co)
    B, _, H, W = x.shape
    arith = split_arith_id()
    wp, aw = conv3x3_sb_pack(weight, transpose_flip, nt)
    if arith and ax is None:
        ax = tensor_amax(x)
    y = torch.empty(B, conv_out, H, W, dtype=F32, device=x.device)
    if addend is not None:
        if tuple(addend.shape) != tuple(y.shape):
            raise RuntimeError("conv3x3_sb_run: addend %s does not have the output's shape %s" % (tuple(addend.shape), tuple(y.shape)))
        _hip.call("cseg_conv3x3_split_fwd_add", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), _p(addend, F32, "addend"), B,
                  conv_in, conv_out, H, W, int(nt), arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y),
                  _hip.stream_ptr())
        return y
    if want_stats and CONV_EPILOGUE_STATS:
        st = tile_stats_buffer(0, conv_out, B, H, W, x.device)
        _hip.call("cseg_conv3x3_split_fwd_st", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), B, conv_in, conv_out, H, W,
                  int(nt), arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y), _pf(st), _hip.stream_ptr())
        return tile_stats_attach(y, st)
    _hip.call("cseg_conv3x3_split_fwd", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), B, conv_in, conv_out, H, W,
              int(nt), arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y), _hip.stream_ptr())
    return y


# Forward / backward-data go to the split-bf16 kernel only when its grid fills the chip (4x64-pixel tiles x channel tiles of
# 144 / 96 / 48); smaller problems stay on MIOpen
# Round 5: the threshold is 1 -- every shape the kernels cover takes them. The old rule (256 blocks, round 2: "one image: 35 vs 25 us on
# MIOpen") priced GPU time only; below ~4 images per GPU the step is HOST-bound, and a torch.conv2d costs the host ~29 us (forward) and
# an aten::convolution_backward ~50 us against ~6 us per call into this library, and MIOpen's output carries no BatchNorm statistics
# (one more pass + launch per site). Measured on the MI355X, ms/step old -> new (profiles/r05_min_tiles.txt): batch 4 62.9 -> 48.9,
# batch 2 57.1 -> 50.5, batch 1 eager 62.0 -> 45.3 (replayed 47.6 -> 42.9), batch 8 85.5 / 87.5 -> 85.6 / 85.8 (unchanged: its layers
# were above the threshold anyway); inside a process group (bench.py --dist-single-rank) batch 4 98.7 -> 83.9, batch 2 79.4 -> 71.8,
# batch 1 79.7 -> 70.2. CSEG_SB_MIN_TILES=256 restores the old routing.
CONV3X3_SB_MIN_TILES = int(os.environ.get("CSEG_SB_MIN_TILES", "1"))


def conv3x3_sb_tiles(x, c_out):
    """Blocks of the launch: with the tiling the call will really use (explicit for CONV3X3_SB_PICK_NT_CHANNELS)."""
    if c_out in CONV3X3_SB_PICK_NT_CHANNELS:
        nt16 = 16 * conv3x3_sb_pick_nt(x, c_out)
    else:
        nt16 = 144 if c_out % 144 == 0 else 96 if c_out % 96 == 0 else 48 if c_out % 48 == 0 else 64      # % 64: four tiles per block
    return x.shape[0] * (c_out // nt16) * ((x.shape[2] + 3) // 4) * ((x.shape[3] + 63) // 64)


# Weight gradient on the split-bf16 kernel (csrc/conv3x3_sb_wrw.hip, version 1), on by default for the channel counts it
# has been measured on: MI355X, bs 8, tools/conv3x3_sb_wrw_probe.py (profiles/r02_conv3x3_split_bf16_wrw_probe.jsonl):
# 48 ch 108 vs 164 us (fp32-MFMA kernel) / 201 (MIOpen), 96 ch 112 vs 152 / 148, 720 ch 17.6 vs 19.5 ms (MIOpen, plus its
# layout transposes); deviation from MIOpen's fp32 result 5e-6 of the gradient scale at those shapes. Parity: 5 shapes vs
# fp64 + determinism on the GPU (tests/test_gpu_conv3x3_sb.py), the same sources on the CPU emulation of the execution
# model incl. the autograd path, and the reference's one-SGD-step golden through the whole HRNet-W48 with this kernel in
# every eligible layer (tools/emu_step_golden.py, profiles/r02_emu_step_golden_*.json). CSEG_CONV3X3_SB_WRW=0 restores the
# fp32-MFMA kernel / MIOpen; CSEG_CONV3X3_SB_WRW_V=2 selects the producer/consumer version (not yet run on hardware).
CONV3X3_SB_WRW = os.environ.get("CSEG_CONV3X3_SB_WRW", "1") == "1"
# 64 / 128 since round 6 (f16x3: output channels % 16, a partly filled last 48-channel block): the layer-1 bottlenecks of HRNet, ResNet's
# layers 1 / 2 -- the last weight gradients of the HRNet step that went to MIOpen's NHWC implicit GEMM and its layout transposes
CONV3X3_SB_WRW_CHANNELS = tuple(int(c) for c in os.environ.get("CSEG_CONV3X3_SB_WRW_CHANNELS", "48,64,96,128,192,384,720").split(","))


# (Cin, Cout) pairs with different channel counts that take the split weight gradient too: transition 1 of HRNet (256 -> 48)
CONV3X3_SB_WRW_PAIRS = ((256, 48),)


def conv3x3_sb_wrw_eligible(x, dy):
    """Shapes the kernel covers (NCHW fp32, Cin % 16, Cout % 48; 64-pixel row segments, 32-pixel ones for the 16 x 32 maps of the
    384-channel branch; since round 5 any width in the f16x3 arithmetic -- the last segment of a row may be ragged)."""
    return (_on_device(x) and x.dtype == F32 and dy.dtype == F32 and x.is_contiguous() and dy.is_contiguous()
            and x.shape[1] % 16 == 0 and (dy.shape[1] % 48 == 0 or (dy.shape[1] % 16 == 0 and SPLIT_ARITH == "f16x3"))
            and (x.shape[3] % 32 == 0 or SPLIT_ARITH == "f16x3"))


def conv3x3_sb_wrw_wanted(x, dy):
    """The autograd path takes the split-bf16 weight gradient: switched on, covered, and a channel count it was timed on."""
    same = x.shape[1] == dy.shape[1] and x.shape[1] in CONV3X3_SB_WRW_CHANNELS
    return CONV3X3_SB_WRW and (same or (x.shape[1], dy.shape[1]) in CONV3X3_SB_WRW_PAIRS) and conv3x3_sb_wrw_eligible(x, dy)


def conv3x3_sb_wrw(x, dy, ax=None, ady=None):
    """dw [Cout,Cin,3,3] of conv2d(x, w, stride 1, padding 1) for the output gradient dy, split-operand MFMA kernel. ax / ady:
    max|x| / max|dy| words when the caller has them (the forward / backward-data calls of the same layer computed both)."""
    B, ci, H, W = x.shape
    co = dy.shape[1]
    lib = _hip.lib()
    n = lib.cseg_conv3x3_sb_wrw_ws_floats(B, ci, co, H, W)
    if n == 0:
        raise RuntimeError("conv3x3_sb_wrw: unsupported shape %s x %s" % (tuple(x.shape), tuple(dy.shape)))
    arith = split_arith_id()
    if arith:
        ax = tensor_amax(x) if ax is None else ax
        ady = tensor_amax(dy) if ady is None else ady
    ws = torch.empty(n, dtype=F32, device=x.device)
    dw = torch.empty(co, ci, 3, 3, dtype=F32, device=x.device)
    _hip.call("cseg_conv3x3_split_wrw", _pq(x, "x"), _pq(dy, "dy"), B, ci, co, H, W, arith,
              _pf(ax) if arith else _null(), _pf(ady) if arith else _null(), _pf(ws), _pf(dw), _hip.stream_ptr())
    return dw


def mark_zero_channel_sum(dx):
    """Called by the BatchNorm backward paths for the input gradient of a BatchNorm that normalised with the BATCH statistics: its sum
    over (batch, pixels) is identically zero per channel -- dx = gamma * invstd * (d - mean(d) - xhat * mean(d * xhat)) and sum(xhat) = 0 --
    (over the whole process group under SyncBN, which is what DDP's gradient average needs). The bias of the convolution in front of
    such a BatchNorm therefore has a zero gradient, and bias_grad() below returns it without a pass over dy (the 720-channel head
    convolutions: 2 x 0.15 ms over 755 MB per step; the reference sums rounding noise there, ~1e-8 of the weight gradients)."""
    if dx is not None:
        dx._cseg_zero_chan_sum = dx._version
    return dx


BIAS_GRAD_SHORTCUT = os.environ.get("CSEG_BIAS_GRAD_SHORTCUT", "1") == "1"


def bias_grad(dy):
    """Gradient of a convolution's bias for the output gradient dy [B, C, H, W]."""
    if BIAS_GRAD_SHORTCUT and getattr(dy, "_cseg_zero_chan_sum", None) == dy._version:
        return torch.zeros(dy.shape[1], dtype=dy.dtype, device=dy.device)
    return dy.sum((0, 2, 3))


class Conv3x3SplitBF16(Function):
    """y = conv2d(x, weight, bias, stride 1, padding 1): forward and backward-data on the split-bf16 MFMA kernel; the
    weight gradient on the split-bf16 kernel too where conv3x3_sb_wrw_wanted() says so (48 / 96 / 720 channels at widths
    that are multiples of 64), else on the fp32-MFMA kernel (bias-free 48/96-channel branches) or on MIOpen together with
    the bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.pick = weight.shape[0] in CONV3X3_SB_PICK_NT_CHANNELS
        ctx.ax = amax_of(x) if split_arith_id() else None              # reused by the weight gradient
        nt = conv3x3_sb_pick_nt(x, weight.shape[0]) if ctx.pick else conv3x3_sb_head_nt(weight.shape[0], x)
        return conv3x3_sb_run(x, weight, False, bias, nt, ax=ctx.ax, want_stats=want_stats)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ady = amax_of(dy) if split_arith_id() else None
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            nt = conv3x3_sb_pick_nt(dy, weight.shape[1]) if ctx.pick else conv3x3_sb_head_nt(weight.shape[1], dy)
            dx = conv3x3_sb_run(dy, weight, True, None, nt, ax=ady)
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            co, ci = weight.shape[:2]
            if conv3x3_sb_wrw_wanted(x, dy):
                dw = _on_wgrad_stream(lambda: conv3x3_sb_wrw(x, dy, ax=ctx.ax, ady=ady), x, dy, ctx.ax, ady) \
                    if ctx.needs_input_grad[1] else None
                db = bias_grad(dy) if want_db else None
            elif not ctx.has_bias and co == ci and co in CONV3X3_WRW_CHANNELS and x.shape[3] % 4 == 0:
                dw = _conv3x3_wrw(x, dy, co, ci)
            else:
                _, dw, db = torch.ops.aten.convolution_backward(
                    dy, x, weight, [co] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                    [False, bool(ctx.needs_input_grad[1]), bool(want_db)])
        return dx, dw, db, None


def conv3x3_split_bf16(x, weight, bias=None, want_stats=False):
    """want_stats: a BatchNorm follows -- let the epilogue produce its statistics (tile_stats_attach on the result)."""
    return Conv3x3SplitBF16.apply(x, weight, bias, want_stats)


class Conv3x3SplitFork(Function):
    """(conv2d(x, weight, None, 1, 1), x): the first convolution of a residual block together with the block's identity path. Forward
    is Conv3x3SplitBF16's; backward receives BOTH gradients that meet at the block input -- dy of the convolution and g of the identity
    path (the masked gradient of the block's last BN + add + ReLU) -- and adds g in the epilogue of the backward-data kernel instead of
    leaving a separate elementwise add to autograd (104 adds of 6-50 MB tensors per step of HRNet-W48: 1.7 ms)."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.pick = weight.shape[0] in CONV3X3_SB_PICK_NT_CHANNELS
        ctx.ax = amax_of(x) if split_arith_id() else None
        nt = conv3x3_sb_pick_nt(x, weight.shape[0]) if ctx.pick else 0
        return conv3x3_sb_run(x, weight, False, None, nt, ax=ctx.ax, want_stats=want_stats), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, g):
        x, weight = ctx.saved_tensors
        ady = amax_of(dy) if split_arith_id() else None
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            nt = conv3x3_sb_pick_nt(dy, weight.shape[1]) if ctx.pick else 0
            dx = conv3x3_sb_run(dy, weight, True, None, nt, ax=ady, addend=None if g is None else g.contiguous())
        dw = None
        if ctx.needs_input_grad[1]:
            co, ci = weight.shape[:2]
            if conv3x3_sb_wrw_wanted(x, dy):
                dw = _on_wgrad_stream(lambda: conv3x3_sb_wrw(x, dy, ax=ctx.ax, ady=ady), x, dy, ctx.ax, ady)
            elif co == ci and co in CONV3X3_WRW_CHANNELS and x.shape[3] % 4 == 0:
                dw = _conv3x3_wrw(x, dy, co, ci)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return dx, dw, None


CONV3X3_FORK = os.environ.get("CSEG_CONV3X3_FORK", "1") == "1"


# ----------------------------------------------------------------------------------------------------------
# A whole residual block as ONE autograd node (round 4). Reference shape of the work: BasicBlock.forward of
# lib/models/backbones/hrnet/hrnet_backbone.py:49-65 (conv3x3 -> bn -> relu -> conv3x3 -> bn -> + x -> relu), 104 of them per
# HRNet-W48 step. The kernels are exactly those of the four nodes it replaces (Conv3x3SplitFork, _BNAct, Conv3x3SplitBF16, _BNAct:
# same calls, same order, same max|.| and statistics hand-overs); what goes away is host work: three of four Function.apply round
# trips and their module __call__ layers per direction -- the batch-8 step had become HOST-bound once the branches ran on forked
# streams (tools/host_profile.py: 91 ms of enqueue time for a ~88 ms step).
# ----------------------------------------------------------------------------------------------------------
BLOCK_FUSED = os.environ.get("CSEG_BLOCK_FUSED", "1") == "1"


def basic_block_split_ok(x, w1, w2):
    """Both convolutions on the split kernels in all three directions (forward / backward-data / weight gradient), as the unfused
    path would route them at this shape."""
    c = w1.shape[0]
    return (BLOCK_FUSED and CONV3X3_FORK and CONV3X3_SPLIT_BF16 and CONV3X3_SB_WRW and _on_device(x) and x.requires_grad
            and tuple(w1.shape) == tuple(w2.shape) == (c, c, 3, 3) and c in CONV3X3_SB_BRANCH_CHANNELS and c in CONV3X3_SB_WRW_CHANNELS
            and conv3x3_sb_eligible(x, w1) and conv3x3_sb_tiles(x, c) >= CONV3X3_SB_MIN_TILES and conv3x3_sb_wrw_eligible(x, x)
            and conv3x3_sb_head_nt(c, x) == 0)


class BasicBlockSplit(Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, bn1, bn2, stats1=True, stats2=True):
        arith = split_arith_id()
        c = w1.shape[0]
        nt = conv3x3_sb_pick_nt(x, c) if c in CONV3X3_SB_PICK_NT_CHANNELS else 0
        ax = amax_of(x) if arith else None
        c1 = conv3x3_sb_run(x, w1, False, None, nt, ax=ax, want_stats=stats1)
        a1, mi1, am1 = _bn_train_fwd(c1, g1, b1, None, bn1)
        c2 = conv3x3_sb_run(a1, w2, False, None, nt, ax=am1, want_stats=stats2)
        out, mi2, am2 = _bn_train_fwd(c2, g2, b2, x, bn2)
        amax_attach(out, am2)
        ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, c1, a1, c2, out, mi1, mi2)
        ctx.misc = (nt, ax, am1)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w1, g1, b1, w2, g2, b2, c1, a1, c2, out, mi1, mi2 = ctx.saved_tensors
        nt, ax, am1 = ctx.misc
        dy = dy.contiguous()
        # bn2 + add + ReLU (mask from `out`; the masked gradient g is also the identity path's gradient)
        am = amax_request(c2)
        dc2, dg2, db2, g = bn_bwd(dy, c2, out, mi2, g2, b2, 2, True, True, amax=am)
        da1 = conv3x3_sb_run(dc2, w2, True, None, nt, ax=am)
        dw2 = _on_wgrad_stream(lambda: conv3x3_sb_wrw(a1, dc2, ax=am1, ady=am), a1, dc2, am1, am) if ctx.needs_input_grad[4] else None
        # bn1 + ReLU (mask recomputed from c1)
        amb = amax_request(c1)
        dc1, dg1, db1, _ = bn_bwd(da1, c1, None, mi1, g1, b1, 1, True, True, amax=amb)
        # conv1: backward-data with the identity path's gradient added in the epilogue
        dx = conv3x3_sb_run(dc1, w1, True, None, nt, ax=amb, addend=g) if ctx.needs_input_grad[0] else None
        dw1 = _on_wgrad_stream(lambda: conv3x3_sb_wrw(x, dc1, ax=ax, ady=amb), x, dc1, ax, amb) if ctx.needs_input_grad[1] else None
        return (dx, dw1, dg1 if g1 is not None else None, db1 if b1 is not None else None, dw2,
                dg2 if g2 is not None else None, db2 if b2 is not None else None, None, None, None, None)


def _bn_train_fwd(x, weight, bias, residual, bn):
    """Single-rank training-mode BN(+residual)+ReLU of lib/models/tools/fused_bn.bn_forward, without its dispatch: statistics from
    the convolution epilogue when it left them, one pass otherwise. -> (y, mean_invstd, max|y| record)."""
    amax = amax_request(x)
    tiles = known_tile_stats(x)
    if tiles is not None:
        mi = bn_tiles_finalize(tiles, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
        y = bn_apply(x, mi, weight, bias, residual, True, amax=amax)
    else:
        y, mi = bn_fwd(x, weight, bias, residual, True, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                       amax=amax)
    return y, mi, amax


def basic_block_split(x, blk):
    """blk: a residual block with conv1 / bn1 / conv2 / bn2 (no downsample, stride 1), BatchNorms in single-rank training mode."""
    return BasicBlockSplit.apply(x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
                                 blk.bn1, blk.bn2, blk.conv1.bn_follows, blk.conv2.bn_follows)


# ----------------------------------------------------------------------------------------------------------
# Round 6: GROUPED launches over the parallel branches of an HRNet exchange unit. Reference shape of the work:
# lib/models/backbones/hrnet/hrnet_backbone.py:262-288 loops over the branches in Python; at every depth of the residual chains the
# n = 2..4 BasicBlocks (:49-65) are independent, of equal flops and of very different shape. Through round 5 every (branch, conv / BN
# pass) was its own launch -- 2 780 dispatches per batch-8 step, 1 664 of them inside these blocks, the coarse branches' launches too
# small for 256 CUs -- with the branches forked onto four HIP streams to get some overlap back. Here ONE autograd node runs a whole
# depth: 6 launches forward (conv group, statistics finalisation, apply; twice), 10 backward, whatever n is; the library's grouped
# entry points (include/cseg_hip.h, ABI 6) schedule the members' tiles inside one kernel. Results per member are those of the one-layer
# calls with the group's tile body (nt = CSEG_NT_GROUP): bit-identical (tests/test_emu_group.py, tools/probes/group_probe.cpp).
# CSEG_BLOCK_GROUP=0 restores the per-branch nodes (and with them the forked streams).
# ----------------------------------------------------------------------------------------------------------
BLOCK_GROUP = os.environ.get("CSEG_BLOCK_GROUP", "1") == "1"
NT_GROUP = _hip.NT_GROUP
_GROUP_SCHED = {}        # (device index, stream) -> scheduling record of the grouped convolution launches (zero between launches)


def _group_sched(device):
    key = (device.index, _hip.raw_stream()) if device.type == "cuda" else (-1, 0)
    buf = _GROUP_SCHED.get(key)
    if buf is None:
        buf = _GROUP_SCHED[key] = torch.zeros(_hip.GROUP_SCHED_INTS, dtype=I32, device=device)       # zero ONCE: every launch leaves it zero
    return buf


def conv3x3_group_run(items, want_stats=False):
    """items: [(x, weight, transpose_flip, ax, addend or None)] -> ([y], [stats or None]): one cseg_conv3x3_split_group_fwd launch.
    y_i = conv2d(x_i, w_i, None, 1, 1) (transpose_flip: the backward-data operator) [+ addend_i]; want_stats: with the BatchNorm
    statistics epilogue (not together with an addend)."""
    n = len(items)
    arr = (_hip.ConvGroupMember * n)()
    ys, sts = [], []
    stats_on = want_stats and CONV_EPILOGUE_STATS
    for e, (x, weight, flip, ax, addend) in zip(arr, items):
        co, ci = weight.shape[:2]
        conv_in, conv_out = (co, ci) if flip else (ci, co)
        B, _, H, W = x.shape
        wp, aw = SPLIT_WEIGHTS.get(weight, "c3", flip, NT_GROUP)
        y = torch.empty(B, conv_out, H, W, dtype=F32, device=x.device)
        e.x, e.wp, e.y, e.amax_x, e.amax_w = _pq(x, "x").value, wp.data_ptr(), y.data_ptr(), ax.data_ptr(), aw.data_ptr()
        if addend is not None:
            e.addend = _pq(addend, "addend").value
        st = None
        if stats_on and addend is None:
            st = tile_stats_buffer(0, conv_out, B, H, W, x.device)
            e.stats = st.data_ptr()
            tile_stats_attach(y, st)
        e.B, e.Cin, e.Cout, e.H, e.W = B, conv_in, conv_out, H, W
        ys.append(y)
        sts.append(st)
    _hip.call("cseg_conv3x3_split_group_fwd", ctypes.byref(arr), n, split_arith_id(), _pf(_group_sched(ys[0].device)), _hip.stream_ptr())
    return ys, sts


def conv3x3_group_wrw(items):
    """items: [(x, dy, ax, ady)] -> [dw]: the weight gradients of the members' convolutions, two launches (cseg_conv3x3_split_group_wrw)."""
    n = len(items)
    arr = (_hip.WrwGroupMember * n)()
    lib = _hip.lib()
    dws, keep = [], []
    for e, (x, dy, ax, ady) in zip(arr, items):
        B, ci, H, W = x.shape
        co = dy.shape[1]
        nf = lib.cseg_conv3x3_sb_wrw_ws_floats(B, ci, co, H, W)
        if nf == 0:
            raise RuntimeError("conv3x3_group_wrw: unsupported shape %s x %s" % (tuple(x.shape), tuple(dy.shape)))
        ws = torch.empty(nf, dtype=F32, device=x.device)
        dw = torch.empty(co, ci, 3, 3, dtype=F32, device=x.device)
        e.x, e.dy, e.amax_x, e.amax_dy, e.ws, e.dw = _pq(x, "x").value, _pq(dy, "dy").value, ax.data_ptr(), ady.data_ptr(), ws.data_ptr(), dw.data_ptr()
        e.B, e.Cin, e.Cout, e.H, e.W = B, ci, co, H, W
        dws.append(dw)
        keep.append(ws)
    _hip.call("cseg_conv3x3_split_group_wrw", ctypes.byref(arr), n, split_arith_id(), _hip.stream_ptr())
    return dws


def bn_group_fwd(xs, bns, residuals, relu):
    """Single-rank training-mode BN(+residual)(+ReLU) of n independent sites whose inputs carry the producing convolution's epilogue
    statistics -> ([y], [mean_invstd], [max|y| record]): cseg_bn_group_tiles_finalize + cseg_bn_group_apply, two launches."""
    n = len(xs)
    arr = (_hip.BnGroupMember * n)()
    ys, mis, ams = [], [], []
    for e, x, bn, r in zip(arr, xs, bns, residuals):
        B, C, HW = _bn_dims(x)
        st = known_tile_stats(x)
        if st is None:
            raise RuntimeError("bn_group_fwd: an input without epilogue statistics")
        mi = torch.empty(C, 2, dtype=F32, device=x.device)
        y = torch.empty_like(x)
        am = amax_request(x)
        p, b = bn._parameters, bn._buffers
        e.x, e.y, e.stats, e.mean_invstd = x.data_ptr(), y.data_ptr(), st.data_ptr(), mi.data_ptr()
        if r is not None:
            e.residual = _pq(r, "residual").value
        w_, b_ = p.get("weight"), p.get("bias")
        if w_ is not None:
            e.weight = w_.data_ptr()
        if b_ is not None:
            e.bias = b_.data_ptr()
        rm, rv, nbt = b.get("running_mean"), b.get("running_var"), b.get("num_batches_tracked")
        if rm is not None:
            e.running_mean, e.running_var = rm.data_ptr(), rv.data_ptr()
        if nbt is not None:
            e.num_batches_tracked = nbt.data_ptr()
        if am is not None:
            e.amax_out = am.data_ptr()
        e.B, e.C, e.HW, e.T, e.eps, e.momentum = B, C, HW, st.shape[1], float(bn.eps), float(bn.momentum)
        ys.append(y)
        mis.append(mi)
        ams.append(am)
    sp = _hip.stream_ptr()
    _hip.call("cseg_bn_group_tiles_finalize", ctypes.byref(arr), n, sp)
    _hip.call("cseg_bn_group_apply", ctypes.byref(arr), n, int(bool(relu)), sp)
    return ys, mis, ams


_BN_GROUP_WS = {}        # (device index, stream) -> reduction scratch of the grouped BN backward, grown on demand


def bn_group_bwd(dys, xs, outs, mis, bns, mode):
    """Adjoint of bn_group_fwd (training statistics, single rank) -> per site (dx, d_weight, d_bias, masked gradient or None, max|dx|
    record): cseg_bn_group_bwd, two launches. mode: 1 = ReLU mask from x, 2 = from `out` (residual sites)."""
    n = len(xs)
    arr = (_hip.BnGroupMember * n)()
    lib = _hip.lib()
    needs, total = [], 0
    for x in xs:
        B, C, HW = _bn_dims(x)
        nf = _BN_WS_NEED.get((B, C, HW))
        if nf is None:
            nf = _BN_WS_NEED[(B, C, HW)] = max(1, lib.cseg_bn_ws_floats(B, C, HW))
        needs.append(total)
        total += (nf + 63) // 64 * 64
    dev = xs[0].device
    key = (dev.index, _hip.raw_stream()) if dev.type == "cuda" else (-1, 0)
    ws = _BN_GROUP_WS.get(key)
    if ws is None or ws.numel() < total:
        ws = _BN_GROUP_WS[key] = torch.empty(max(total, 1 << 17), dtype=F32, device=dev)
    base = ws.data_ptr()
    res = []
    for e, dy, x, out, mi, bn, off in zip(arr, dys, xs, outs, mis, bns, needs):
        B, C, HW = _bn_dims(x)
        d_wb = torch.empty(2, C, dtype=F32, device=dev)
        g = torch.empty_like(x) if mode == 2 else None
        dx = torch.empty_like(x)
        am = amax_request(x)
        p = bn._parameters
        e.dy, e.x, e.mean_invstd, e.dx, e.ws = _pq(dy, "dy").value, x.data_ptr(), mi.data_ptr(), dx.data_ptr(), base + 4 * off
        e.d_weight, e.d_bias = d_wb.data_ptr(), d_wb.data_ptr() + 4 * C
        if mode == 2:
            e.out, e.g_masked = out.data_ptr(), g.data_ptr()
        w_, b_ = p.get("weight"), p.get("bias")
        if w_ is not None:
            e.weight = w_.data_ptr()
        if b_ is not None:
            e.bias = b_.data_ptr()
        if am is not None:
            e.amax_out = am.data_ptr()
        e.B, e.C, e.HW = B, C, HW
        res.append((dx, d_wb[0], d_wb[1], g, am))
    _hip.call("cseg_bn_group_bwd", ctypes.byref(arr), n, int(mode), 1, _hip.stream_ptr())
    return res


# ---- SyncBN forms of the grouped BatchNorm passes: the statistics of a depth cross the ranks as ONE packed fp64 tensor between two
# launches (lib/models/tools/fused_bn.BasicBlockGroupSync issues the all-reduce)
def bn_group_sync_moments(xs):
    """xs: convolution outputs that carry epilogue statistics -> packed moments [sum (C_i + 1), 2] f64 (row C_i of a member = this rank's
    element count): cseg_bn_group_tiles_moments, one launch."""
    n = len(xs)
    arr = (_hip.BnGroupMember * n)()
    rows = 0
    for e, x in zip(arr, xs):
        B, C, HW = _bn_dims(x)
        st = known_tile_stats(x)
        if st is None:
            raise RuntimeError("bn_group_sync_moments: an input without epilogue statistics")
        e.stats, e.B, e.C, e.HW, e.T = st.data_ptr(), B, C, HW, st.shape[1]
        rows += C + 1
    packed = torch.empty(rows, 2, dtype=F64, device=xs[0].device)
    _hip.call("cseg_bn_group_tiles_moments", ctypes.byref(arr), n, _pf(packed), _hip.stream_ptr())
    return packed


def bn_group_sync_apply(xs, bns, residuals, relu, packed):
    """The all-reduced packed moments -> mean / invstd / running statistics per site (cseg_bn_group_finalize), then
    y = act(bn(x) [+ residual]) for all sites (cseg_bn_group_apply): two launches -> ([y], [mean_invstd], [max|y| record])."""
    n = len(xs)
    arr = (_hip.BnGroupMember * n)()
    ys, mis, ams = [], [], []
    for e, x, bn, r in zip(arr, xs, bns, residuals):
        B, C, HW = _bn_dims(x)
        mi = torch.empty(C, 2, dtype=F32, device=x.device)
        y = torch.empty_like(x)
        am = amax_request(x)
        p, b = bn._parameters, bn._buffers
        e.x, e.y, e.mean_invstd = x.data_ptr(), y.data_ptr(), mi.data_ptr()
        if r is not None:
            e.residual = _pq(r, "residual").value
        w_, b_ = p.get("weight"), p.get("bias")
        if w_ is not None:
            e.weight = w_.data_ptr()
        if b_ is not None:
            e.bias = b_.data_ptr()
        rm, rv, nbt = b.get("running_mean"), b.get("running_var"), b.get("num_batches_tracked")
        if rm is not None:
            e.running_mean, e.running_var = rm.data_ptr(), rv.data_ptr()
        if nbt is not None:
            e.num_batches_tracked = nbt.data_ptr()
        if am is not None:
            e.amax_out = am.data_ptr()
        e.B, e.C, e.HW, e.eps, e.momentum = B, C, HW, float(bn.eps), float(bn.momentum)
        ys.append(y)
        mis.append(mi)
        ams.append(am)
    sp = _hip.stream_ptr()
    _hip.call("cseg_bn_group_finalize", ctypes.byref(arr), n, _pf(packed), sp)
    _hip.call("cseg_bn_group_apply", ctypes.byref(arr), n, int(bool(relu)), sp)
    return ys, mis, ams


def bn_group_sync_bwd_reduce(dys, xs, outs, mis, bns, mode):
    """First half of the grouped SyncBN backward: masked gradients (mode 2), rank-local d_weight / d_bias and the packed fp64 gradient
    sums [sum (C_i + 1), 2] the host all-reduces (cseg_bn_group_bwd_reduce, two launches) -> (packed, state for bn_group_sync_bwd_apply)."""
    n = len(xs)
    arr = (_hip.BnGroupMember * n)()
    lib = _hip.lib()
    needs, total, rows = [], 0, 0
    for x in xs:
        B, C, HW = _bn_dims(x)
        nf = _BN_WS_NEED.get((B, C, HW))
        if nf is None:
            nf = _BN_WS_NEED[(B, C, HW)] = max(1, lib.cseg_bn_ws_floats(B, C, HW))
        needs.append(total)
        total += (nf + 63) // 64 * 64
        rows += C + 1
    dev = xs[0].device
    key = (dev.index, _hip.raw_stream()) if dev.type == "cuda" else (-1, 0)
    ws = _BN_GROUP_WS.get(key)
    if ws is None or ws.numel() < total:
        ws = _BN_GROUP_WS[key] = torch.empty(max(total, 1 << 17), dtype=F32, device=dev)
    base = ws.data_ptr()
    packed = torch.empty(rows, 2, dtype=F64, device=dev)
    res = []
    for e, dy, x, out, mi, bn, off in zip(arr, dys, xs, outs, mis, bns, needs):
        B, C, HW = _bn_dims(x)
        d_wb = torch.empty(2, C, dtype=F32, device=dev)
        g = torch.empty_like(x) if mode == 2 else None
        p = bn._parameters
        e.dy, e.x, e.mean_invstd, e.ws = _pq(dy, "dy").value, x.data_ptr(), mi.data_ptr(), base + 4 * off
        e.d_weight, e.d_bias = d_wb.data_ptr(), d_wb.data_ptr() + 4 * C
        if mode == 2:
            e.out, e.g_masked = out.data_ptr(), g.data_ptr()
        w_, b_ = p.get("weight"), p.get("bias")
        if w_ is not None:
            e.weight = w_.data_ptr()
        if b_ is not None:
            e.bias = b_.data_ptr()
        e.B, e.C, e.HW = B, C, HW
        res.append([None, d_wb[0], d_wb[1], g, None])
    _hip.call("cseg_bn_group_bwd_reduce", ctypes.byref(arr), n, int(mode), _pf(packed), _hip.stream_ptr())
    return packed, (arr, res, mode, xs, dys)


def bn_group_sync_bwd_apply(state, packed):
    """Second half: dx of every site from the all-reduced packed sums (cseg_bn_group_bwd_apply, one launch) -> per site
    (dx, d_weight, d_bias, masked gradient or None, max|dx| record)."""
    arr, res, mode, xs, _dys = state
    for e, x, r in zip(arr, xs, res):
        dx = torch.empty_like(x)
        am = amax_request(x)
        e.dx = dx.data_ptr()
        if am is not None:
            e.amax_out = am.data_ptr()
        r[0], r[4] = dx, am
    _hip.call("cseg_bn_group_bwd_apply", ctypes.byref(arr), len(xs), int(mode), _pf(packed), _hip.stream_ptr())
    return [tuple(r) for r in res]


def _group_wrw(items):
    """conv3x3_group_wrw on the weight-gradient stream when wgrad_scope is open (Trainer.train_step, CSEG_WGRAD_STREAM=1): the two
    launches of a depth's weight gradients then run beside the BatchNorm passes and the next backward-data launch of the chain instead
    of between them (the gradients feed nothing until the optimizer runs)."""
    if not _WGRAD["on"] or not items[0][0].is_cuda or torch.cuda.is_current_stream_capturing():
        return conv3x3_group_wrw(items)
    cur = torch.cuda.current_stream(items[0][0].device)
    side = _WGRAD["stream"]
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        dws = conv3x3_group_wrw(items)
    for it in items:
        for t in it:
            t.record_stream(side)
    for dw in dws:
        dw.record_stream(_WGRAD["main"])             # read by the optimizer on the trainer's stream, after wgrad_scope joined the side stream
    _WGRAD["used"] = True
    return dws


def basic_block_group_ok(blocks, xs):
    """Every block of the depth qualifies for the grouped node: what BasicBlockSplit needs (split kernels in all three directions,
    plain single-rank batch statistics from the convolution epilogues), f16x3, at least two members."""
    if not (BLOCK_GROUP and 2 <= len(blocks) <= _hip.GROUP_MAX and split_arith_id() == ARITH_IDS["f16x3"] and CONV_EPILOGUE_STATS):
        return False
    for blk, x in zip(blocks, xs):
        c = blk.conv1.weight.shape[0]
        if not (c >= 32 and blk.conv1.bn_follows and blk.conv2.bn_follows and blk.bn1.momentum is not None and blk.bn2.momentum is not None
                and basic_block_split_ok(x, blk.conv1.weight, blk.conv2.weight)):
            return False
    return True


class BasicBlockGroup(Function):
    """n residual blocks (conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> + x -> ReLU) of one depth of parallel branches as ONE node on
    the grouped launches. tensors: per block x, w1, g1, b1, w2, g2, b2. Same arithmetic per block as BasicBlockSplit."""

    @staticmethod
    def forward(ctx, blocks, *tensors):
        n = len(blocks)
        xs = [tensors[7 * i].contiguous() for i in range(n)]
        w1s = [tensors[7 * i + 1] for i in range(n)]
        w2s = [tensors[7 * i + 4] for i in range(n)]
        axs = [amax_of(x) for x in xs]
        c1s, _ = conv3x3_group_run([(x, w, False, ax, None) for x, w, ax in zip(xs, w1s, axs)], want_stats=True)
        a1s, mi1, am1 = bn_group_fwd(c1s, [b.bn1 for b in blocks], [None] * n, True)
        c2s, _ = conv3x3_group_run([(a, w, False, am, None) for a, w, am in zip(a1s, w2s, am1)], want_stats=True)
        outs, mi2, am2 = bn_group_fwd(c2s, [b.bn2 for b in blocks], xs, True)
        for o, am in zip(outs, am2):
            amax_attach(o, am)
        ctx.blocks, ctx.axs, ctx.am1 = blocks, axs, am1
        ctx.save_for_backward(*(xs + c1s + a1s + c2s + outs + mi1 + mi2))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        blocks, axs, am1 = ctx.blocks, ctx.axs, ctx.am1
        n = len(blocks)
        sv = ctx.saved_tensors
        xs, c1s, a1s, c2s, outs, mi1, mi2 = (list(sv[k * n:(k + 1) * n]) for k in range(7))
        need = ctx.needs_input_grad
        dys = [d.contiguous() for d in dys]
        # bn2 + add + ReLU (mask from `out`; the masked gradient g is also the identity path's gradient)
        r2 = bn_group_bwd(dys, c2s, outs, mi2, [b.bn2 for b in blocks], 2)
        da1s, _ = conv3x3_group_run([(r2[i][0], blocks[i].conv2.weight, True, r2[i][4], None) for i in range(n)])
        dw2s = _group_wrw([(a1s[i], r2[i][0], am1[i], r2[i][4]) for i in range(n)])
        # bn1 + ReLU (mask recomputed from c1)
        r1 = bn_group_bwd(da1s, c1s, [None] * n, mi1, [b.bn1 for b in blocks], 1)
        # conv1: backward-data with the identity path's gradient added in the epilogue
        dxs, _ = conv3x3_group_run([(r1[i][0], blocks[i].conv1.weight, True, r1[i][4], r2[i][3]) for i in range(n)])
        dw1s = _group_wrw([(xs[i], r1[i][0], axs[i], r1[i][4]) for i in range(n)])
        grads = [None]
        for i, blk in enumerate(blocks):
            _, dg1, db1, _, _ = r1[i]
            _, dg2, db2, _, _ = r2[i]
            grads += [dxs[i] if need[1 + 7 * i] else None, dw1s[i] if need[2 + 7 * i] else None,
                      dg1 if blk.bn1.weight is not None else None, db1 if blk.bn1.bias is not None else None,
                      dw2s[i] if need[5 + 7 * i] else None,
                      dg2 if blk.bn2.weight is not None else None, db2 if blk.bn2.bias is not None else None]
        return tuple(grads)


def basic_block_group(blocks, xs):
    """The residual blocks of one depth of parallel branches on inputs xs -> outputs, as ONE node on the grouped launches, or None
    when a block does not qualify (the caller then runs the branches one by one)."""
    if not basic_block_group_ok(blocks, xs):
        return None
    tensors = []
    for blk, x in zip(blocks, xs):
        tensors += [x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
    return list(BasicBlockGroup.apply(tuple(blocks), *tensors))


# ----------------------------------------------------------------------------------------------------------
# Dilated 3x3 convolutions (round 5, csrc/conv3x3_sb16.hip: conv3x3_sb16d_kernel): rate 2 / 4, padding = rate, stride 1 -- layer3 / layer4
# of the dilated ResNet encoders of DeepLab-V3 (reference lib/models/backbones/resnet/resnet_backbone.py:88-101). Forward and
# backward-data on the split kernel (f16x3), weight / bias gradients on MIOpen.
# ----------------------------------------------------------------------------------------------------------
CONV3X3_DIL_SPLIT = os.environ.get("CSEG_CONV3X3_DIL_SPLIT", "1") == "1"


def conv3x3_dil_eligible(x, weight, dilation):
    """NCHW fp32 on the GPU, 3x3, rate 2 or 4, channel counts that are packed in the 16-channel-chunk form both ways (multiples of 64
    that are not multiples of 48: 64, 128, 256, 512, 1024, ...), f16x3 arithmetic."""
    if not (CONV3X3_DIL_SPLIT and CONV3X3_SPLIT_BF16 and SPLIT_ARITH == "f16x3" and _on_device(x) and x.dtype == F32 and weight.dtype == F32
            and x.dim() == 4 and x.is_contiguous()):
        return False
    co, ci, kh, kw = weight.shape
    ok = lambda c: c % 64 == 0 and c % 48 != 0
    return (kh, kw) == (3, 3) and tuple(dilation) in ((2, 2), (4, 4)) and ok(ci) and ok(co) and x.shape[1] == ci


def conv3x3_dil_run(x, weight, dil, transpose_flip=False, bias=None, ax=None, addend=None, want_stats=False):
    """y = conv2d(x, weight, bias, stride 1, padding dil, dilation dil) (transpose_flip: the backward-data operator applied to x)."""
    co, ci = weight.shape[:2]
    conv_in, conv_out = (co, ci) if transpose_flip else (ci, co)
    B, _, H, W = x.shape
    wp, aw = SPLIT_WEIGHTS.get(weight, "c3", transpose_flip, 0)
    if ax is None:
        ax = tensor_amax(x)
    y = torch.empty(B, conv_out, H, W, dtype=F32, device=x.device)
    st = tile_stats_buffer(0, conv_out, B, H, W, x.device) if (want_stats and CONV_EPILOGUE_STATS and addend is None) else None
    _hip.call("cseg_conv3x3_split_dil_fwd", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"),
              _p(addend, F32, "addend") if addend is not None else _null(), B, conv_in, conv_out, H, W, int(dil), split_arith_id(),
              _pf(ax), _pf(aw), _pf(y), _pf(st) if st is not None else _null(), _hip.stream_ptr())
    return tile_stats_attach(y, st) if st is not None else y


class Conv3x3DilSplit(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dil, want_stats=False):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.dil = bias is not None, int(dil)
        ctx.ax = amax_of(x)
        return conv3x3_dil_run(x, weight, dil, False, bias, ax=ctx.ax, want_stats=want_stats)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d = ctx.dil
        ady = amax_of(dy)
        dy = dy.contiguous()
        dx = conv3x3_dil_run(dy, weight, d, True, None, ax=ady) if ctx.needs_input_grad[0] else None
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            _, dw, db = torch.ops.aten.convolution_backward(
                dy, x, weight, [weight.shape[0]] if ctx.has_bias else None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
                [False, bool(ctx.needs_input_grad[1]), bool(want_db)])
        return dx, dw, db, None, None


def conv3x3_dil_split(x, weight, bias, dil, want_stats=False):
    return Conv3x3DilSplit.apply(x, weight, bias, dil, want_stats)


def conv3x3_split_fork(x, weight, want_stats=False):
    return Conv3x3SplitFork.apply(x, weight, want_stats)


# ----------------------------------------------------------------------------------------------------------
# 3x3 / stride 2 / pad 1 on the split kernels (csrc/conv3x3_s2.hip, csrc/conv3x3_sb_wrw.hip): f16x3 only
# ----------------------------------------------------------------------------------------------------------
# HRNet's 52 downsampling convolutions (fuse + transition layers): on MIOpen 32 + 33 Winograd launches and ~50 NHWC implicit-GEMM
# weight gradients with their layout transposes per step, ~15 ms of the 108 ms step (profiles/r03_step_steady_kernel_stats.csv).
# Each direction is routed on its own (256 -> 96 has no 48-multiple on the input side: its backward-data stays on MIOpen).
CONV3X3_S2_SPLIT = os.environ.get("CSEG_CONV3X3_S2_SPLIT", "1") == "1"


def _s2_base_ok(x, weight):
    # CONV3X3_SPLIT_BF16 off = the strict fp32 configuration (bench.py --conv-arith fp32): nothing on the 16-bit matrix pipes
    return (CONV3X3_S2_SPLIT and CONV3X3_SPLIT_BF16 and SPLIT_ARITH == "f16x3" and _on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def conv3x3_s2_fwd_eligible(x, weight):
    co, ci = weight.shape[:2]
    return _s2_base_ok(x, weight) and x.shape[1] == ci and ci % 16 == 0 and (co % 48 == 0 or co % 64 == 0)      # (even input; any output width since round 5; 64: the stem's second convolution, round 6)


def conv3x3_s2_bwd_eligible(x, weight):
    co, ci = weight.shape[:2]
    return _s2_base_ok(x, weight) and co % 16 == 0 and (ci % 48 == 0 or ci % 64 == 0) and x.shape[3] % 4 == 0


def conv3x3_s2_wrw_eligible(x, weight):
    co, ci = weight.shape[:2]
    return _s2_base_ok(x, weight) and ci % 16 == 0 and co % 16 == 0 and x.shape[3] % 64 == 0


def conv3x3_s2_pick_nt(B, Ho, Wo, c_out):
    """16-channel tiles per block: 6 when the grid still has >= 256 blocks of 4 x 64 outputs, else 3."""
    spatial = B * ((Ho + 3) // 4) * ((Wo + 63) // 64)
    if c_out % 48:
        return 4                                         # 256 channels (input side of transition 1)
    return 6 if c_out % 96 == 0 and spatial * (c_out // 96) >= 256 else 3


def conv3x3_s2_run(x, weight, ax=None, want_stats=False):
    """y = conv2d(x, weight, None, stride 2, padding 1)."""
    co, ci = weight.shape[:2]
    B, _, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    nt = conv3x3_s2_pick_nt(B, Ho, Wo, co)
    wp, aw = SPLIT_WEIGHTS.get(weight, "c3s2", False, nt)
    ax = tensor_amax(x) if ax is None else ax
    y = torch.empty(B, co, Ho, Wo, dtype=F32, device=x.device)
    if want_stats and CONV_EPILOGUE_STATS:
        st = tile_stats_buffer(0, co, B, Ho, Wo, x.device)
        _hip.call("cseg_conv3x3_s2_split_fwd_st", _p(x, F32, "x"), wp.data_ptr(), B, ci, co, Ho, Wo, nt, _pf(ax), _pf(aw), _pf(y),
                  _pf(st), _hip.stream_ptr())
        return tile_stats_attach(y, st)
    _hip.call("cseg_conv3x3_s2_split_fwd", _p(x, F32, "x"), wp.data_ptr(), B, ci, co, Ho, Wo, nt, _pf(ax), _pf(aw), _pf(y),
              _hip.stream_ptr())
    return y


def conv3x3_s2_bwd_run(dy, weight, ady=None):
    """dx of that convolution for the output gradient dy [B, Cout, Ho, Wo] -> [B, Cin, 2 Ho, 2 Wo]."""
    co, ci = weight.shape[:2]
    B, _, Ho, Wo = dy.shape
    nt = conv3x3_s2_pick_nt(B, Ho, Wo, ci)
    wp, aw = SPLIT_WEIGHTS.get(weight, "c3s2", True, nt)
    ady = tensor_amax(dy) if ady is None else ady
    dx = torch.empty(B, ci, 2 * Ho, 2 * Wo, dtype=F32, device=dy.device)
    _hip.call("cseg_conv3x3_s2_split_bwd", _p(dy, F32, "dy"), wp.data_ptr(), B, ci, co, Ho, Wo, nt, _pf(ady), _pf(aw), _pf(dx),
              _hip.stream_ptr())
    return dx


def conv3x3_s2_wrw(x, dy, ax=None, ady=None):
    """dw [Cout, Cin, 3, 3] of that convolution."""
    B, ci, H, W = x.shape
    co, Ho, Wo = dy.shape[1:]
    n = _hip.lib().cseg_conv3x3_s2_wrw_ws_floats(B, ci, co, Ho, Wo)
    if n == 0 or (H, W) != (2 * Ho, 2 * Wo):
        raise RuntimeError("conv3x3_s2_wrw: unsupported shape %s x %s" % (tuple(x.shape), tuple(dy.shape)))
    ax = tensor_amax(x) if ax is None else ax
    ady = tensor_amax(dy) if ady is None else ady
    ws = torch.empty(n, dtype=F32, device=x.device)
    dw = torch.empty(co, ci, 3, 3, dtype=F32, device=x.device)
    _hip.call("cseg_conv3x3_s2_split_wrw", _p(x, F32, "x"), _p(dy, F32, "dy"), B, ci, co, Ho, Wo, split_arith_id(), _pf(ax), _pf(ady),
              _pf(ws), _pf(dw), _hip.stream_ptr())
    return dw


class Conv3x3S2Split(Function):
    """y = conv2d(x, weight, None, stride 2, padding 1) with every direction the split kernels cover on them and the others on
    MIOpen (aten)."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        weight = weight.contiguous()
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.ax = amax_of(x)                              # reused by the weight gradient
        if conv3x3_s2_fwd_eligible(x, weight):
            return conv3x3_s2_run(x, weight, ax=ctx.ax, want_stats=want_stats)
        return torch.nn.functional.conv2d(x, weight, None, 2, 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        co, ci = weight.shape[:2]
        ady = amax_of(dy)
        dy = dy.contiguous()
        need_dx, need_dw = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])
        dx = dw = None
        if need_dx and conv3x3_s2_bwd_eligible(x, weight):
            dx = conv3x3_s2_bwd_run(dy, weight, ady=ady)
        if need_dw and conv3x3_s2_wrw_eligible(x, weight):
            dw = _on_wgrad_stream(lambda: conv3x3_s2_wrw(x, dy, ax=ctx.ax, ady=ady), x, dy, ctx.ax, ady)
        rest = [need_dx and dx is None, need_dw and dw is None, False]
        if rest[0] or rest[1]:
            gx, gw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, rest)
            dx = gx if rest[0] else dx
            dw = gw if rest[1] else dw
        return dx, dw, None


def conv3x3_s2_split(x, weight, want_stats=False):
    return Conv3x3S2Split.apply(x, weight, want_stats)


# ----------------------------------------------------------------------------------------------------------
# The first convolution of the stem, nn.Conv2d(3, 64, 3, 2, 1, bias=False) on the image (csrc/conv3x3_stem.hip, round 6): fp32 FMA
# kernels for the forward and the weight gradient -- the image needs no gradient, so there is no backward-data operator
# ----------------------------------------------------------------------------------------------------------
CONV3X3_RGB_STEM = os.environ.get("CSEG_CONV3X3_RGB_STEM", "1") == "1"


def conv3x3_s2_rgb_eligible(x, weight):
    return (CONV3X3_RGB_STEM and _on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4 and x.shape[1] == 3
            and tuple(weight.shape) == (64, 3, 3, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and not x.requires_grad)


class Conv3x3S2Rgb(Function):
    """y = conv2d(x, weight, None, stride 2, padding 1) for x [B, 3, H, W] that needs no gradient and weight [64, 3, 3, 3]."""

    @staticmethod
    def forward(ctx, x, weight):
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        B, _, H, W = x.shape
        y = torch.empty(B, 64, H // 2, W // 2, dtype=F32, device=x.device)
        _hip.call("cseg_conv3x3_s2_rgb_fwd", _p(x, F32, "x"), _p(weight, F32, "weight"), B, 64, H, W, _pf(y), _hip.stream_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None
        dy = dy.contiguous()
        B, _, H, W = x.shape
        n = _hip.lib().cseg_conv3x3_s2_rgb_wrw_ws_floats(B, 64, H, W)
        if n == 0:
            raise RuntimeError("conv3x3_s2_rgb_wrw: unsupported shape %s" % (tuple(x.shape),))

        def wrw():
            ws = torch.empty(n, dtype=F32, device=x.device)
            dw = torch.empty(64, 3, 3, 3, dtype=F32, device=x.device)
            _hip.call("cseg_conv3x3_s2_rgb_wrw", _p(x, F32, "x"), _p(dy, F32, "dy"), B, 64, H, W, _pf(ws), _pf(dw), _hip.stream_ptr())
            return dw
        return None, _on_wgrad_stream(wrw, x, dy)


def conv3x3_s2_rgb(x, weight):
    return Conv3x3S2Rgb.apply(x, weight)


# ----------------------------------------------------------------------------------------------------------
# The classifier convolution (csrc/cls1x1.hip, round 6): 1x1 onto K <= 32 output channels from a wide activation, with the channel
# dropout in front of it folded into per-image weights. Reference: the tail of `cls_head`, lib/models/nets/hrnet.py:73-80
# (BNReLU -> nn.Dropout2d(0.10) -> nn.Conv2d(720, num_classes, 1)); there a dropout pass over the 755 MB activation forward and
# backward, and a library GEMM with 19 columns each way (rocBLAS 0.40 + 0.26 ms, MIOpen's NHWC weight gradient + two transposes 0.54 ms).
# ----------------------------------------------------------------------------------------------------------
CLS1X1 = os.environ.get("CSEG_CLS1X1", "1") == "1"
CLS1X1_MAX_K = 32


def cls1x1_eligible(x, weight):
    """NCHW fp32 on the GPU, a 1x1 kernel with at most 32 output channels."""
    return (CLS1X1 and _on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4 and weight.dim() == 4
            and tuple(weight.shape[2:]) == (1, 1) and weight.shape[0] <= CLS1X1_MAX_K and weight.shape[1] == x.shape[1])


def cls1x1_kp(k):
    return 20 if k <= 20 else 32


class Cls1x1(Function):
    """y [B,K,H,W] = bias + sum_c wt[b][c][k] x[b][c]; wt [B, C, KP] (kernels.cls1x1_weights) carries the transposed, padded and
    channel-masked weights, and receives its gradient in the same layout (autograd takes it back to the [K, C, 1, 1] parameter)."""

    @staticmethod
    def forward(ctx, x, wt, bias, K_):
        x, wt = x.contiguous(), wt.contiguous()
        B, C, H, W = x.shape
        KP = wt.shape[2]
        y = torch.empty(B, K_, H, W, dtype=F32, device=x.device)
        _hip.call("cseg_cls1x1_fwd", _p(x, F32, "x"), _p(wt, F32, "wt"), _opt(bias, F32, "bias"), B, C, K_, KP, ctypes.c_long(H * W),
                  _pf(y), _hip.stream_ptr())
        ctx.save_for_backward(x, wt)
        ctx.K = K_
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        dy = dy.contiguous()
        B, C, H, W = x.shape
        K_, KP, P = ctx.K, wt.shape[2], H * W
        dx = dwt = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _hip.call("cseg_cls1x1_bwd", _p(dy, F32, "dy"), _p(wt, F32, "wt"), B, C, K_, KP, ctypes.c_long(P), _pf(dx), _hip.stream_ptr())
        if ctx.needs_input_grad[1]:
            n = _hip.lib().cseg_cls1x1_wrw_ws_floats(B, C, KP, ctypes.c_long(P))
            if n == 0:
                raise RuntimeError("cls1x1_wrw: unsupported shape %s x %s" % (tuple(x.shape), tuple(dy.shape)))
            ws = torch.empty(n, dtype=F32, device=x.device)
            dwt = torch.empty(B, C, KP, dtype=F32, device=x.device)
            _hip.call("cseg_cls1x1_wrw", _p(x, F32, "x"), _p(dy, F32, "dy"), B, C, K_, KP, ctypes.c_long(P), _pf(ws), _pf(dwt),
                      _hip.stream_ptr())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dwt, db, None


def cls1x1_weights(weight, B, mask=None):
    """[K, C, 1, 1] parameter -> wt [B, C, KP]: transposed, zero-padded to KP columns, times the channel mask [B, C] of a folded
    Dropout2d (entries 0 or 1 / (1 - p)) when there is one. A few small tensor operations on 19 x 720 numbers, differentiable."""
    K_, C = weight.shape[:2]
    KP = cls1x1_kp(K_)
    w2 = weight.reshape(K_, C).t()
    if KP > K_:
        w2 = torch.nn.functional.pad(w2, (0, KP - K_))
    if mask is None:
        return w2.unsqueeze(0).expand(B, C, KP).contiguous()
    return w2.unsqueeze(0) * mask.reshape(B, C, 1)


def cls1x1(x, weight, bias=None, mask=None):
    return Cls1x1.apply(x, cls1x1_weights(weight, x.shape[0], mask), bias, weight.shape[0])


# ----------------------------------------------------------------------------------------------------------
# 1x1 convolution on the BF16 matrix cores with split operands (csrc/conv1x1_sb.hip): first hardware run pending -> opt-in
# ----------------------------------------------------------------------------------------------------------
CONV1X1_SPLIT_BF16 = os.environ.get("CSEG_CONV1X1_SPLIT_BF16", "1") == "1"
CONV1X1_SB_MIN_TILES = int(os.environ.get("CSEG_SB_MIN_TILES", "1"))       # (see CONV3X3_SB_MIN_TILES)


def conv1x1_sb_eligible(x, weight):
    """NCHW fp32 on the GPU, 1x1, Cin % 16 and Cout % 48|64 in both directions (backward-data swaps them)."""
    if not (_on_device(x) and x.dtype == F32 and weight.dtype == F32 and x.dim() == 4 and x.is_contiguous()):
        return False
    co, ci, kh, kw = weight.shape
    ok = lambda c: c % 48 == 0 or c % 64 == 0
    return (kh, kw) == (1, 1) and ok(ci) and ok(co) and x.shape[1] == ci          # (any H*W since round 5, see conv3x3_sb_eligible)


def conv1x1_sb_tiles(x, c_out):
    nt16 = next(16 * nt for nt in (9, 8, 6, 4, 3) if c_out % (16 * nt) == 0)
    return x.shape[0] * (c_out // nt16) * ((x.shape[2] * x.shape[3] + 255) // 256)


def conv1x1_sb_pack(weight, transpose=False):
    return SPLIT_WEIGHTS.get(weight, "c1", transpose, 0)


def conv1x1_sb_run(x, weight, transpose=False, bias=None, ax=None, want_stats=False, addend=None):
    """y = conv2d(x, weight[Cout,Cin,1,1], bias) (transpose: the backward-data operator applied to x); addend: a tensor of y's shape added
    in the epilogue (not together with want_stats)."""
    co, ci = weight.shape[:2]
    conv_in, conv_out = (co, ci) if transpose else (ci, co)
    B, _, H, W = x.shape
    arith = split_arith_id()
    wp, aw = conv1x1_sb_pack(weight, transpose)
    if arith and ax is None:
        ax = tensor_amax(x)
    y = torch.empty(B, conv_out, H, W, dtype=F32, device=x.device)
    if want_stats and CONV_EPILOGUE_STATS:
        st = tile_stats_buffer(1, conv_out, B, H * W, 1, x.device)
        _hip.call("cseg_conv1x1_split_fwd_st", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), B, conv_in, conv_out, H * W,
                  arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y), _pf(st), _hip.stream_ptr())
        return tile_stats_attach(y, st)
    if addend is not None:
        _hip.call("cseg_conv1x1_split_fwd_add", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), _pq(addend, "addend"), B, conv_in,
                  conv_out, H * W, arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y), _hip.stream_ptr())
        return y
    _hip.call("cseg_conv1x1_split_fwd", _pq(x, "x"), wp.data_ptr(), _opt(bias, F32, "bias"), B, conv_in, conv_out, H * W,
              arith, _pf(ax) if arith else _null(), _pf(aw) if arith else _null(), _pf(y), _hip.stream_ptr())
    return y


# weight gradient on the split kernel where it won in the round-2 driver pass (720x720: 2.98 vs 5.75 ms, 720x256: 0.96 vs 1.31 ms
# on MIOpen; the 64 <-> 256 bottleneck convolutions lost: 0.30 vs 0.22 ms) -> both channel counts >= CONV1X1_SB_WRW_MIN_CH
CONV1X1_SB_WRW = os.environ.get("CSEG_CONV1X1_SB_WRW", "1") == "1"
CONV1X1_SB_WRW_MIN_CH = int(os.environ.get("CSEG_CONV1X1_SB_WRW_MIN_CH", "16"))      # round 3, lean loader: 98.2 ms/step at 256, 97.6 at 48, 96.8 at 16 (same box)


def conv1x1_sb_wrw_eligible(x, dy):
    return (_on_device(x) and x.dtype == F32 and dy.dtype == F32 and x.is_contiguous() and dy.is_contiguous()
            and x.shape[1] % 16 == 0 and dy.shape[1] % 16 == 0)          # (any H*W since round 5: ragged loader variant)


def conv1x1_sb_wrw_wanted(x, dy):
    return (CONV1X1_SB_WRW and min(x.shape[1], dy.shape[1]) >= CONV1X1_SB_WRW_MIN_CH and conv1x1_sb_wrw_eligible(x, dy))


def conv1x1_sb_wrw(x, dy, ax=None, ady=None):
    """dw [Cout,Cin,1,1] of a 1x1 convolution for the output gradient dy, split-operand MFMA kernel."""
    B, ci, H, W = x.shape
    co = dy.shape[1]
    lib = _hip.lib()
    n = lib.cseg_conv1x1_sb_wrw_ws_floats(B, ci, co, H * W)
    if n == 0:
        raise RuntimeError("conv1x1_sb_wrw: unsupported shape %s x %s" % (tuple(x.shape), tuple(dy.shape)))
    arith = split_arith_id()
    if arith:
        ax = tensor_amax(x) if ax is None else ax
        ady = tensor_amax(dy) if ady is None else ady
    ws = torch.empty(n, dtype=F32, device=x.device)
    dw = torch.empty(co, ci, 1, 1, dtype=F32, device=x.device)
    _hip.call("cseg_conv1x1_split_wrw", _pq(x, "x"), _pq(dy, "dy"), B, ci, co, H * W, arith,
              _pf(ax) if arith else _null(), _pf(ady) if arith else _null(), _pf(ws), _pf(dw), _hip.stream_ptr())
    return dw


class Conv1x1SplitBF16(Function):
    """y = conv2d(x, weight, bias) for a 1x1 kernel: forward and backward-data on the split-bf16 MFMA kernel, weight / bias
    gradients on MIOpen / rocBLAS (fp32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.ax = amax_of(x) if split_arith_id() else None
        return conv1x1_sb_run(x, weight, False, bias, ax=ctx.ax, want_stats=want_stats)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ady = amax_of(dy) if split_arith_id() else None
        dy = dy.contiguous()
        dx = conv1x1_sb_run(dy, weight, True, ax=ady) if ctx.needs_input_grad[0] else None
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if conv1x1_sb_wrw_wanted(x, dy):
            dw = _on_wgrad_stream(lambda: conv1x1_sb_wrw(x, dy, ax=ctx.ax, ady=ady), x, dy, ctx.ax, ady) \
                if ctx.needs_input_grad[1] else None
            db = bias_grad(dy) if want_db else None
        elif ctx.needs_input_grad[1] or want_db:
            _, dw, db = torch.ops.aten.convolution_backward(
                dy, x, weight, [weight.shape[0]] if ctx.has_bias else None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                [False, bool(ctx.needs_input_grad[1]), bool(want_db)])
        return dx, dw, db, None


def conv1x1_split_bf16(x, weight, bias=None, want_stats=False):
    return Conv1x1SplitBF16.apply(x, weight, bias, want_stats)


class Conv1x1SplitSkip(Function):
    """(conv2d(x, weight, bias), x) -- the first 1x1 convolution of a residual block together with the block's skip connection
    (reference lib/models/backbones/hrnet/hrnet_backbone.py:68-105: `residual = x ... out += residual`). Forward: the convolution of
    Conv1x1SplitBF16 and an alias of x for the skip path. Backward: the two gradients that meet at x -- W^T dy and what comes back over
    the skip -- in ONE pass: the backward-data kernel adds the skip gradient in its epilogue (cseg_conv1x1_split_fwd_add) instead of
    autograd's separate add over two 268 MB tensors per block of layer 1."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        weight = weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.ax = amax_of(x) if split_arith_id() else None
        return conv1x1_sb_run(x, weight, False, bias, ax=ctx.ax, want_stats=want_stats), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, weight = ctx.saved_tensors
        ady = amax_of(dy) if split_arith_id() else None
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            add = None
            if dskip is not None and dskip.dtype == F32 and dskip.shape == x.shape and _on_device(dskip):
                add = dskip.contiguous()
            dx = conv1x1_sb_run(dy, weight, True, ax=ady, addend=add)
            if dskip is not None and add is None:
                dx = dx + dskip
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if conv1x1_sb_wrw_wanted(x, dy):
            dw = _on_wgrad_stream(lambda: conv1x1_sb_wrw(x, dy, ax=ctx.ax, ady=ady), x, dy, ctx.ax, ady) \
                if ctx.needs_input_grad[1] else None
            db = bias_grad(dy) if want_db else None
        elif ctx.needs_input_grad[1] or want_db:
            _, dw, db = torch.ops.aten.convolution_backward(
                dy, x, weight, [weight.shape[0]] if ctx.has_bias else None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                [False, bool(ctx.needs_input_grad[1]), bool(want_db)])
        return dx, dw, db, None


SKIP_ADD_FUSED = os.environ.get("CSEG_SKIP_ADD_FUSED", "1") == "1"


def conv1x1_split_skip(x, weight, bias=None, want_stats=False):
    """-> (y, skip): see Conv1x1SplitSkip. The max|.| record of x travels with the alias."""
    y, skip = Conv1x1SplitSkip.apply(x, weight, bias, want_stats)
    a = getattr(x, "_cseg_amax", None)
    if a is not None and a[1] == x._version:
        skip._cseg_amax = (a[0], skip._version) + tuple(a[2:])
    return y, skip


def bn_fwd(x, weight, bias, residual, relu, eps, momentum, running_mean, running_var, num_batches_tracked, amax=None):
    """Single-rank training forward (2 launches): -> (y, mean_invstd [C,2]). amax: zeroed word that receives max|y|."""
    B, C, HW = _bn_dims(x)
    mi = torch.empty(C, 2, dtype=F32, device=x.device)
    y = torch.empty_like(x)
    _hip.call("cseg_bn_fwd_amax", _pq(x, "x"), _pq(residual, "residual") if residual is not None else _null(), _opt(weight, F32, "weight"),
              _opt(bias, F32, "bias"), int(bool(relu)), B, C, HW, _pf(_bn_ws(B, C, HW, x.device)), float(eps),
              float(momentum), _opt(running_mean, F32, "running_mean"), _opt(running_var, F32, "running_var"),
              _opt(num_batches_tracked, I64, "num_batches_tracked"), _pf(mi), _pf(y),
              _pf(amax) if amax is not None else _null(), _hip.stream_ptr())
    return y, mi


def bn_bwd(dy, x, out, mean_invstd, weight, bias, mode, training, want_dx, amax=None):
    """Single-rank backward (2 launches): -> (dx or None, d_weight, d_bias, g_masked or None). amax: zeroed word for max|dx|."""
    B, C, HW = _bn_dims(x)
    dev = x.device
    d_wb = torch.empty(2, C, dtype=F32, device=dev)
    d_weight, d_bias = d_wb[0], d_wb[1]
    g = torch.empty_like(x) if mode == 2 else None
    dx = torch.empty_like(x) if want_dx else None
    _hip.call("cseg_bn_bwd_amax", _pq(dy, "dy"), _pf(x), _pf(out) if out is not None else _null(),
              _pf(mean_invstd), _opt(weight, F32, "weight"), _opt(bias, F32, "bias"), int(mode),
              int(bool(training)), B, C, HW, _pf(_bn_ws(B, C, HW, dev)), _pf(g) if g is not None else _null(),
              _pf(d_weight), _pf(d_bias), _pf(dx) if dx is not None else _null(),
              _pf(amax) if (amax is not None and dx is not None) else _null(), _hip.stream_ptr())
    return dx, d_weight, d_bias, g


# ----------------------------------------------------------------------------------------------------------
# GPU data pipeline (csrc/augment.hip); host logic: lib/datasets/tools/gpu_aug.py
# ----------------------------------------------------------------------------------------------------------
AUG_PARAM_INTS = 12


@torch.no_grad()
def augment_batch(img_u8, lab_u8, lut, params, out_hw, div_value, mean, std):
    """img_u8 [B,Hs,Ws,3] u8, lab_u8 [B,Hs,Ws] u8 or None, lut i16 [256] or None, params i32 [B,AUG_PARAM_INTS]
    (host tensor is uploaded) -> (img f32 [B,3,Ht,Wt], labelmap i64 [B,Ht,Wt] or None)."""
    B, Hs, Ws, _ = img_u8.shape
    Ht, Wt = out_hw
    dev = img_u8.device
    params = params.to(dev, non_blocking=True)
    out = torch.empty(B, 3, Ht, Wt, dtype=F32, device=dev)
    out_lab = torch.empty(B, Ht, Wt, dtype=I64, device=dev) if lab_u8 is not None else None
    mean3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    std3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    U8 = torch.uint8
    _hip.call("cseg_augment_batch", _p(img_u8, U8, "img"), _opt(lab_u8, U8, "labelmap"), _opt(lut, I16, "lut"),
              _p(params, I32, "params"), B, Hs, Ws, Ht, Wt, float(div_value), mean3, std3, _p(out, F32, "out_img"),
              _opt(out_lab, I64, "out_lab"), _hip.stream_ptr())
    return out, out_lab
