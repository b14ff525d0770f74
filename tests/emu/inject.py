"""TEST INFRASTRUCTURE. Points the product's ctypes binding (contrastiveseg_amd/_hip.py) at libcseg_emu.so -- the kernel
sources compiled for the host against the CPU emulation of the execution model -- so that the autograd wrappers of
contrastiveseg_amd/kernels.py and everything above them run unchanged on CPU tensors, with the HIP SOURCES doing the
device work. Done by tests only (monkeypatch); the product itself has no such switch and refuses CPU tensors."""
import ctypes

import torch

from . import build_emu


def install(monkeypatch):
    from contrastiveseg_amd import _hip
    try:
        path = build_emu.build()
    except build_emu.EmuBuildError as e:               # an environment without the host toolchain: not a test failure
        import pytest
        pytest.skip(str(e))
    handle = ctypes.CDLL(path)
    for name, (res, args) in _hip.SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args

    def dev(t, dtype, what):
        if t.is_cuda:
            raise RuntimeError("%s: the emulated library takes host tensors" % what)
        if t.dtype != dtype:
            raise RuntimeError("%s must be %s (got %s)" % (what, dtype, t.dtype))
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % what)
        return ctypes.c_void_p(t.data_ptr())

    from contrastiveseg_amd import kernels
    monkeypatch.setattr(kernels, "_on_device", lambda t: not t.is_cuda)
    monkeypatch.setattr(_hip, "_lib", handle)
    monkeypatch.setattr(_hip, "dev", dev)
    monkeypatch.setattr(_hip, "stream_ptr", lambda: ctypes.c_void_p(None))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return handle
