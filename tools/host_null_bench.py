"""Host-side cost of the hot Python paths WITHOUT a GPU: the ctypes binding is pointed at a stub whose compute entry points return
at once (size queries go to the real library), tensors live on the CPU and are never touched. What is timed is what the enqueueing
thread pays per launch -- Python, autograd nodes, allocations, argument conversion -- i.e. the part of tools/host_profile.py's
"host enqueue" that is not the HIP runtime. Absolute numbers are this container's CPU; the ratios guide the work.
  python tools/host_null_bench.py [block|module] [--profile]"""
import cProfile
import ctypes
import io
import os
import pstats
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from contrastiveseg_amd import _hip
from contrastiveseg_amd import kernels as K

QUERIES = ("_bytes", "_floats", "_segments", "_plan", "abi_version", "last_error", "_words")


def install_null():
    d = tempfile.mkdtemp()
    src = os.path.join(d, "stub.c")
    open(src, "w").write("int cseg_stub(void) { return 1; }\n")
    so = os.path.join(d, "libstub.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-o", so, src])
    stub = ctypes.CDLL(so)
    real = _hip.lib()

    class Null:
        pass
    null = Null()
    for name, (res, args) in _hip.SIGNATURES.items():
        if any(q in name for q in QUERIES):
            setattr(null, name, getattr(real, name))
        else:
            fn = ctypes.CFUNCTYPE(res, *args)(("cseg_stub", stub))          # same argument conversion as the real entry point
            setattr(null, name, fn)
    _hip._lib = null
    _hip.dev = lambda t, dtype, what: ctypes.c_void_p(t.data_ptr())
    _hip.stream_ptr = lambda: ctypes.c_void_p(None)
    _hip.current_stream_id = lambda: 0
    K._on_device = lambda t: True
    torch.cuda.synchronize = lambda *a, **k: None


def bench_block(profile):
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    torch.manual_seed(0)
    n_blk = 8                                     # a chain, so that the engine's per-call cost is shared like in a real backward
    blk = torch.nn.Sequential(*[BasicBlock(48, 48, bn_type="torchbn") for _ in range(n_blk)]).train()
    mark_conv_bn_pairs(blk)
    # small tensors: the stub kernels do not care, and real CPU work on 50 MB tensors (the `* 1.0` below, its gradient, mmap-backed
    # allocations) would drown what is measured
    K.CONV3X3_SB_MIN_TILES = 1
    x = torch.zeros(2, 48, 8, 64).requires_grad_(True)
    dy = torch.zeros(2, 48, 8, 64)

    def run(n):
        t_f = t_b = 0.0
        for _ in range(n):
            xin = x * 1.0
            t0 = time.perf_counter()
            y = blk(xin)
            t1 = time.perf_counter()
            y.backward(dy)
            t2 = time.perf_counter()
            t_f += t1 - t0
            t_b += t2 - t1
        return t_f / n * 1e6, t_b / n * 1e6
    run(5)
    f, b = run(50)
    print("BasicBlock 48 ch: forward %.0f us, backward %.0f us of host time per block (chain of %d, stub kernels)" % (f / n_blk, b / n_blk, n_blk))
    if profile:
        pr = cProfile.Profile()
        pr.enable()
        run(50)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
        print(s.getvalue())


def bench_module(profile):
    """One stage-4 exchange unit of HRNet-W48 (4 branches x 4 residual blocks + the 12 fuse paths) at the benched relative sizes,
    scaled down 4 x (the stub kernels do not care; the aten fallbacks of shapes the routing refuses stay small)."""
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import HighResolutionModule
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    K.CONV3X3_SB_MIN_TILES = 1
    torch.manual_seed(0)
    chans = [48, 96, 192, 384]
    mod = mark_conv_bn_pairs(HighResolutionModule(chans, 4, "torchbn", 0.1).train())
    xs0 = [torch.zeros(2, c, 64 >> i, 128 >> i) for i, c in enumerate(chans)]
    dys = [torch.zeros_like(t) for t in xs0]
    calls = []
    orig = _hip.call
    _hip.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]

    def run(n):
        t_f = t_b = 0.0
        for _ in range(n):
            xs = [t.clone().requires_grad_(True) for t in xs0]
            t0 = time.perf_counter()
            ys = mod([t * 1.0 for t in xs])
            t1 = time.perf_counter()
            torch.autograd.backward(ys, dys)
            t2 = time.perf_counter()
            t_f += t1 - t0
            t_b += t2 - t1
        return t_f / n * 1e3, t_b / n * 1e3
    run(2)
    calls.clear()
    run(1)
    n_calls = len(calls)
    _hip.call = orig
    f, b = run(10)
    print("stage-4 exchange unit: forward %.2f ms, backward %.2f ms of host time (%d library calls per forward + backward%s)"
          % (f, b, n_calls, ""))
    if profile:
        pr = cProfile.Profile()
        pr.enable()
        run(5)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
        print(s.getvalue())


if __name__ == "__main__":
    install_null()
    (bench_module if "module" in sys.argv else bench_block)("--profile" in sys.argv)
