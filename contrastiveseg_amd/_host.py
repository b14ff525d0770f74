"""ctypes binding of libcseg_host.so, the native host-side helper (csrc_host/rng_draws.cpp): batched replicas of the
`torch.randperm` calls of the reference on PyTorch's default CPU generator. Optional: when the library has not been
built the callers fall back to calling `torch.randperm` once per draw -- the same stream, only slower."""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (libtorch must be loaded before the helper)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcseg_host.so")
_lib = None
_tried = False


def lib():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.path.exists(LIB_PATH) and not os.environ.get("CSEG_NO_HOST_LIB"):
            try:
                h = ctypes.CDLL(LIB_PATH)
                h.cseg_host_randperm_prefixes.restype = ctypes.c_int
                h.cseg_host_randperm_prefixes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                          ctypes.c_void_p]
                _lib = h
            except OSError:
                _lib = None
    return _lib


def randperm_prefixes(n_list, keep, flat=False):
    """For every i: the first keep[i] entries of torch.randperm(n_list[i]), drawn in order from the default CPU
    generator (n_list[i]-1 draws each, also when keep[i] == 0). Returns a list of int64 arrays (flat: ONE array, the prefixes
    back to back)."""
    n_list = np.ascontiguousarray(n_list, dtype=np.int64)
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    h = lib()
    if h is None:
        parts = [torch.randperm(int(n)).numpy()[:int(k)] for n, k in zip(n_list, keep)]
        return (np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)) if flat else parts
    out = np.empty(int(keep.sum()), dtype=np.int64)
    ok = h.cseg_host_randperm_prefixes(n_list.ctypes.data_as(ctypes.c_void_p), keep.ctypes.data_as(ctypes.c_void_p),
                                       len(n_list), out.ctypes.data_as(ctypes.c_void_p))
    if ok != 1:
        raise RuntimeError("cseg_host_randperm_prefixes failed")
    if flat:
        return out
    return np.split(out, np.cumsum(keep)[:-1]) if len(keep) else []
