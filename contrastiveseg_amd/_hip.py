"""ctypes binding of libcseg_hip.so (the C-ABI declared in include/cseg_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every call below passes raw device pointers and
the current HIP stream. There is NO CPU fallback: a missing library or a non-GPU tensor raises (the reference's
native ops fail the same way, lib/extensions/inplace_abn/functions.py:25-28 `_check`)."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcseg_hip.so")

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_ptr = ctypes.c_void_p


class ContrastDesc(ctypes.Structure):
    _fields_ = [("mode", _c_int), ("N", _c_int), ("M", _c_int), ("D", _c_int),
                ("anchors", _ptr), ("a_lab", _ptr), ("contrast", _ptr), ("c_lab", _ptr),
                ("segment_queue", _ptr), ("pixel_queue", _ptr),
                ("bank_classes", _c_int), ("bank_size", _c_int),
                ("temperature", _c_float), ("base_temperature", _c_float)]


class ConvGroupMember(ctypes.Structure):
    """cseg_conv_group_member of include/cseg_hip.h: one convolution of a grouped launch."""
    _fields_ = [("x", _ptr), ("wp", _ptr), ("bias", _ptr), ("addend", _ptr), ("y", _ptr), ("stats", _ptr), ("amax_x", _ptr),
                ("amax_w", _ptr), ("B", _c_int), ("Cin", _c_int), ("Cout", _c_int), ("H", _c_int), ("W", _c_int),
                ("reserved", _c_int * 3)]


class BnGroupMember(ctypes.Structure):
    """cseg_bn_group_member of include/cseg_hip.h: one BatchNorm site of a grouped launch."""
    _fields_ = [("x", _ptr), ("residual", _ptr), ("y", _ptr), ("stats", _ptr), ("mean_invstd", _ptr), ("weight", _ptr), ("bias", _ptr),
                ("running_mean", _ptr), ("running_var", _ptr), ("num_batches_tracked", _ptr), ("amax_out", _ptr), ("dy", _ptr),
                ("out", _ptr), ("g_masked", _ptr), ("d_weight", _ptr), ("d_bias", _ptr), ("dx", _ptr), ("ws", _ptr),
                ("B", _c_int), ("C", _c_int), ("HW", _c_int), ("T", ctypes.c_long), ("eps", _c_float), ("momentum", _c_float)]


class WrwGroupMember(ctypes.Structure):
    """cseg_wrw_group_member of include/cseg_hip.h: one weight gradient of a grouped launch."""
    _fields_ = [("x", _ptr), ("dy", _ptr), ("amax_x", _ptr), ("amax_dy", _ptr), ("ws", _ptr), ("dw", _ptr), ("B", _c_int), ("Cin", _c_int),
                ("Cout", _c_int), ("H", _c_int), ("W", _c_int), ("reserved", _c_int * 3)]


GROUP_MAX = 8              # CSEG_GROUP_MAX
GROUP_SCHED_INTS = 320     # CSEG_GROUP_SCHED_INTS
NT_GROUP = 0x203           # CSEG_NT_GROUP


# name -> (restype, argtypes); must list every symbol of include/cseg_hip.h (tests/test_cabi.py checks it)
SIGNATURES = {
    "cseg_abi_version": (_c_int, []),
    "cseg_last_error": (ctypes.c_char_p, []),
    "cseg_classify_partition": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 7 + [_ptr] * 7 + [_ptr]),
    "cseg_gather_anchors": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr]),
    "cseg_scatter_anchor_grad": (_c_int, [_ptr, _c_int, _ptr, _c_int, _c_int, _c_int, _c_float, _ptr, _ptr]),
    "cseg_contrast_ws_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_contrast_fwd": (_c_int, [ctypes.POINTER(ContrastDesc), _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_contrast_fused_ws_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_contrast_fused_counter_offset": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_contrast_fwd_fused": (_c_int, [ctypes.POINTER(ContrastDesc), _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_contrast_bwd_parts": (_c_int, [_c_int, _c_int, _c_int]),
    "cseg_contrast_bwd": (_c_int, [ctypes.POINTER(ContrastDesc), _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_upcat_fwd": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr]),
    "cseg_upcat_fwd_amax": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_upcat_bwd": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr]),
    "cseg_fuse_sum_fwd": (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _c_int] + [_c_int] * 5 + [_ptr, _ptr]),
    "cseg_affine_channels": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, ctypes.c_long, _ptr, _ptr, _ptr]),
    "cseg_fuse_sum_fwd_amax": (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _c_int] + [_c_int] * 5 + [_ptr, _ptr, _ptr]),
    "cseg_fuse_sum_bwd": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int] + [_c_int] * 4 + [_ptr, _ptr, _ptr]),
    "cseg_upsample_ce_blocks": (_c_int, [_c_int, _c_int, _c_int]),
    "cseg_upsample_ce_fwd": (_c_int, [_ptr, _ptr, _ptr, _c_int] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_upsample_ce_bwd": (_c_int, [_ptr, _ptr, _ptr, _c_int] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_queue_count": (_c_int, [_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_queue_class_sums": (_c_int, [_ptr, _ptr] + [_c_int] * 7 + [_ptr, _ptr]),
    "cseg_queue_write_segments": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _ptr]),
    "cseg_queue_write_pixels": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _c_int,
                                         _ptr]),
    "cseg_conv3x3_packed_floats": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_conv3x3_pack_weights": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_fwd": (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_wrw_ws_floats": (ctypes.c_size_t, [_c_int] * 5),
    "cseg_conv3x3_wrw": (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_sb_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_conv3x3_sb_pack_weights": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_sb_fwd": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_sb_pack_weights_nt": (_c_int, [_ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_sb_fwd_nt": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_sb_wrw_ws_floats": (ctypes.c_size_t, [_c_int] * 5),
    "cseg_conv3x3_sb_wrw": (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_sb_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "cseg_conv1x1_sb_pack_weights": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv1x1_sb_fwd": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv1x1_sb_wrw_ws_floats": (ctypes.c_size_t, [_c_int] * 4),
    "cseg_conv1x1_sb_wrw": (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_amax_f32": (_c_int, [_ptr, ctypes.c_long, _ptr, _ptr]),
    "cseg_conv3x3_split_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int]),
    "cseg_conv3x3_split_pack": (_c_int, [_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_split_fwd": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 7 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_split_wrw": (_c_int, [_ptr, _ptr] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_split_fwd_add": (_c_int, [_ptr, _ptr, _ptr, _ptr] + [_c_int] * 7 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_split_group_fwd": (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_split_group_wrw": (_c_int, [_ptr, _c_int, _c_int, _ptr]),
    "cseg_bn_group_tiles_finalize": (_c_int, [_ptr, _c_int, _ptr]),
    "cseg_bn_group_apply": (_c_int, [_ptr, _c_int, _c_int, _ptr]),
    "cseg_bn_group_bwd": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr]),
    "cseg_bn_group_tiles_moments": (_c_int, [_ptr, _c_int, _ptr, _ptr]),
    "cseg_bn_group_finalize": (_c_int, [_ptr, _c_int, _ptr, _ptr]),
    "cseg_bn_group_bwd_reduce": (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr]),
    "cseg_bn_group_bwd_apply": (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr]),
    "cseg_conv3x3_split_dil_fwd": (_c_int, [_ptr, _ptr, _ptr, _ptr] + [_c_int] * 7 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_split_packed_bytes": (ctypes.c_size_t, [_c_int] * 2),
    "cseg_conv3x3_s2_split_plan": (_c_int, [_c_int] * 4 + [_ptr, _ptr]),
    "cseg_conv3x3_s2_split_pack": (_c_int, [_ptr] + [_c_int] * 4 + [_ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_split_fwd": (_c_int, [_ptr, _ptr] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_split_bwd": (_c_int, [_ptr, _ptr] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_rgb_fwd": (_c_int, [_ptr, _ptr] + [_c_int] * 4 + [_ptr, _ptr]),
    "cseg_conv3x3_s2_rgb_wrw_ws_floats": (ctypes.c_size_t, [_c_int] * 4),
    "cseg_conv3x3_s2_rgb_wrw": (_c_int, [_ptr, _ptr] + [_c_int] * 4 + [_ptr, _ptr, _ptr]),
    "cseg_cls1x1_fwd": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 4 + [ctypes.c_long, _ptr, _ptr]),
    "cseg_cls1x1_bwd": (_c_int, [_ptr, _ptr] + [_c_int] * 4 + [ctypes.c_long, _ptr, _ptr]),
    "cseg_cls1x1_wrw_ws_floats": (ctypes.c_size_t, [_c_int, _c_int, _c_int, ctypes.c_long]),
    "cseg_cls1x1_wrw": (_c_int, [_ptr, _ptr] + [_c_int] * 4 + [ctypes.c_long, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_wrw_ws_floats": (ctypes.c_size_t, [_c_int] * 5),
    "cseg_conv3x3_s2_split_wrw": (_c_int, [_ptr, _ptr] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int]),
    "cseg_conv1x1_split_pack": (_c_int, [_ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_fwd": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 5 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_fwd_add": (_c_int, [_ptr, _ptr, _ptr, _ptr] + [_c_int] * 5 + [_ptr, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_wrw": (_c_int, [_ptr, _ptr] + [_c_int] * 5 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_split_plan": (_c_int, [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_plan": (_c_int, [_c_int, _c_int, _ptr, _ptr]),
    "cseg_conv1x1_split_plan_arith": (_c_int, [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_long)]),
    "cseg_amax_batch": (_c_int, [_ptr, _c_int, _c_int, _ptr]),
    "cseg_split_pack_batch": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr]),
    "cseg_augment_batch": (_c_int, [_ptr, _ptr, _ptr, _ptr] + [_c_int] * 5 + [_c_float, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv_stat_segments": (ctypes.c_size_t, [_c_int] * 4),
    "cseg_conv3x3_split_fwd_st": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 7 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv1x1_split_fwd_st": (_c_int, [_ptr, _ptr, _ptr] + [_c_int] * 5 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_conv3x3_s2_split_fwd_st": (_c_int, [_ptr, _ptr] + [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_bn_tiles_finalize": (_c_int, [_ptr, _c_int, ctypes.c_long, _c_float, _c_float, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_bn_tiles_moments": (_c_int, [_ptr, _c_int, ctypes.c_long, _ptr, _ptr]),
    "cseg_bn_ws_floats": (ctypes.c_size_t, [_c_int, _c_int, _c_int]),
    "cseg_bn_stats": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_bn_finalize": (_c_int, [_ptr, _c_int, ctypes.c_double, _c_float, _c_float, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "cseg_bn_stats_finalize": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _c_float, _c_float, _ptr, _ptr, _ptr, _ptr,
                                        _ptr]),
    "cseg_bn_fwd": (_c_int, [_ptr] * 4 + [_c_int] * 4 + [_ptr, _c_float, _c_float] + [_ptr] * 5 + [_ptr]),
    "cseg_bn_bwd": (_c_int, [_ptr] * 6 + [_c_int] * 5 + [_ptr] * 5 + [_ptr]),
    "cseg_bn_fwd_amax": (_c_int, [_ptr] * 4 + [_c_int] * 4 + [_ptr, _c_float, _c_float] + [_ptr] * 5 + [_ptr, _ptr]),
    "cseg_bn_bwd_amax": (_c_int, [_ptr] * 6 + [_c_int] * 5 + [_ptr] * 5 + [_ptr, _ptr]),
    "cseg_bn_apply_amax": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_bn_bwd_apply_amax": (_c_int, [_ptr] * 6 + [ctypes.c_double, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "cseg_bn_apply": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "cseg_bn_bwd_reduce": (_c_int, [_ptr] * 6 + [_c_int] * 4 + [_ptr] * 5 + [_ptr]),
    "cseg_bn_bwd_apply": (_c_int, [_ptr] * 6 + [ctypes.c_double, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
}

_lib = None


def lib():
    """Loads the library once. Raises if it was not built (run `python __graft_entry__.py` / csrc/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcseg_hip.so is missing at %s: the HIP extension was not built; there is no CPU "
                               "fallback for the contrastive hot path" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(ok, what):
    if ok != 1:
        raise RuntimeError("%s failed: %s" % (what, lib().cseg_last_error().decode()))


_raw_device = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def raw_stream():
    """Current HIP stream of the current device as an integer (key of per-stream scratch buffers)."""
    return _raw_stream(_raw_device())


def stream_ptr():
    """The current HIP stream of the CURRENT device: `dev()` below insists that every tensor lives on that device, so
    kernel, stream and pointers always agree."""
    # the raw accessors: torch.cuda.current_stream() builds a Stream object and goes through three Python layers (7.7 us per call,
    # ~1000 calls per step: 7 ms of the host's 79 ms per step at batch 8, tools/host_profile.py)
    return ctypes.c_void_p(_raw_stream(_raw_device()))


def dev(t, dtype, what):
    """Validates a tensor that is about to be handed to the kernels as a raw pointer."""
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s): libcseg_hip has no CPU path" % (what, t.device))
    if t.device.index != _raw_device():
        raise RuntimeError("%s lives on %s but the current device is cuda:%d: call torch.cuda.set_device(local_rank) "
                           "(one process per GPU) or wrap the call in torch.cuda.device(tensor.device)"
                           % (what, t.device, torch.cuda.current_device()))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % what)
    return ctypes.c_void_p(t.data_ptr())


def call(name, *args):
    _check(getattr(lib(), name)(*args), name)
