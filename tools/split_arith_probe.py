"""bf16x6 vs f16x3 on the MI355X at the benched shapes: per-launch time (HIP events over a loop; weights packed once, amax words
computed once -- what a training step pays per call) and error against an fp64 convolution of the same operands, with MIOpen's
fp32 kernel as the yardstick. One JSON line per (operator, shape). Usage: split_arith_probe.py [fwd] [wrw] [c1]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch
import torch.nn.functional as F

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / iters
        best = t if best is None else min(best, t)
    return round(best, 1)


def err(a, ref):
    return float((a.double() - ref).abs().max())


def with_arith(name, fn):
    old = K.SPLIT_ARITH
    K.SPLIT_ARITH = name
    try:
        return fn()
    finally:
        K.SPLIT_ARITH = old


FWD = {"head_720": (8, 720, 128, 256), "branch_48": (8, 48, 128, 256), "branch_96": (8, 96, 64, 128), "branch_192": (8, 192, 32, 64),
       "branch_384": (8, 384, 16, 32)}


def fwd():
    for name, (B, C, H, W) in FWD.items():
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
        nt = K.conv3x3_sb_pick_nt(x, C) if C in K.CONV3X3_SB_PICK_NT_CHANNELS else 0
        ref = F.conv2d(x[:1].double().cpu(), w.double().cpu(), None, 1, 1).to(dev) if C <= 192 else None
        if ref is None:          # the head: fp64 on the GPU for one image (rocBLAS fp64 GEMM path of aten)
            ref = F.conv2d(x[:1].double(), w.double(), None, 1, 1)
        row = {"op": "conv3x3 fwd", "shape": name, "B": B, "C": C, "HW": [H, W], "nt": nt}
        iters = 5 if C == 720 else 30
        for ar in ("bf16x6", "f16x3"):
            def run():
                ax = K.tensor_amax(x) if ar == "f16x3" else None
                t = timeit(lambda: K.conv3x3_sb_run(x, w, False, None, nt, ax=ax), iters)
                t_amax = timeit(lambda: K.tensor_amax(x), iters) if ar == "f16x3" else 0.0
                e = err(K.conv3x3_sb_run(x[:1].contiguous(), w, False, None, nt), ref)
                return t, t_amax, e
            t, ta, e = with_arith(ar, run)
            row[ar] = {"us": t, "amax_us": ta, "max_err_vs_fp64": e}
        row["miopen_fp32"] = {"us": timeit(lambda: F.conv2d(x, w, None, 1, 1), iters),
                              "max_err_vs_fp64": err(F.conv2d(x[:1], w, None, 1, 1), ref)}
        row["out_absmax"] = float(ref.abs().max())
        print(json.dumps(row), flush=True)


def wrw():
    for name, (B, C, H, W) in FWD.items():
        if W % 64:
            continue
        g = torch.Generator().manual_seed(2)
        x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
        dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
        ref = torch.nn.grad.conv2d_weight(x.double(), (C, C, 3, 3), dy.double(), padding=1)
        row = {"op": "conv3x3 wrw", "shape": name, "B": B, "C": C, "HW": [H, W]}
        iters = 5 if C == 720 else 30
        for ar in ("bf16x6", "f16x3"):
            def run():
                ax, ad = (K.tensor_amax(x), K.tensor_amax(dy)) if ar == "f16x3" else (None, None)
                t = timeit(lambda: K.conv3x3_sb_wrw(x, dy, ax=ax, ady=ad), iters)
                return t, err(K.conv3x3_sb_wrw(x, dy), ref)
            t, e = with_arith(ar, run)
            row[ar] = {"us": t, "max_err_vs_fp64": e}
        wt = torch.zeros(C, C, 3, 3, device=dev)
        mi = lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        row["miopen_fp32"] = {"us": timeit(mi, iters), "max_err_vs_fp64": err(mi(), ref)}
        row["out_absmax"] = float(ref.abs().max())
        print(json.dumps(row), flush=True)


def c1():
    for ci, co in ((720, 720), (720, 256), (64, 256), (256, 64)):
        B, H, W = 8, 128, 256
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, ci, H, W, generator=g).relu_().to(dev)
        w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(dev)
        dy = (torch.randn(B, co, H, W, generator=g) * 1e-3).to(dev)
        ref = F.conv2d(x[:1].double(), w.double())
        refw = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 1, 1), dy.double())
        row = {"op": "conv1x1", "shape": "%d_%d" % (ci, co)}
        for ar in ("bf16x6", "f16x3"):
            def run():
                ax, ad = (K.tensor_amax(x), K.tensor_amax(dy)) if ar == "f16x3" else (None, None)
                return {"fwd_us": timeit(lambda: K.conv1x1_sb_run(x, w, False, None, ax=ax), 10),
                        "bwd_us": timeit(lambda: K.conv1x1_sb_run(dy, w, True, None, ax=ad), 10),
                        "wrw_us": timeit(lambda: K.conv1x1_sb_wrw(x, dy, ax=ax, ady=ad), 10),
                        "fwd_err": err(K.conv1x1_sb_run(x[:1].contiguous(), w), ref), "wrw_err": err(K.conv1x1_sb_wrw(x, dy), refw)}
            row[ar] = with_arith(ar, run)
        wt = torch.zeros_like(w)
        mi = lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        row["torch_fp32"] = {"fwd_us": timeit(lambda: F.conv2d(x, w), 10), "wrw_us": timeit(mi, 10),
                             "fwd_err": err(F.conv2d(x[:1], w), ref), "wrw_err": err(mi(), refw)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fwd", "wrw", "c1"]
    for k in what:
        {"fwd": fwd, "wrw": wrw, "c1": c1}[k]()
