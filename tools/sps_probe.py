"""Weights staged one K-step (CSEG_CONV3X3_SB_SPS=1) or one filter row (default, 3 K-steps) at a time in the f16x3 3x3 forward
kernel: time of the head convolution (720 -> 720, 8x128x256), of its data gradient (same kernel, flipped weights) and of the
96-channel branch, plus the largest deviation between the two forms (they must agree bit for bit: same MFMA order)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
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


g = torch.Generator().manual_seed(1)
for (B, C, H, W) in ((8, 720, 128, 256), (8, 96, 64, 128), (8, 48, 128, 256)):
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
    ax = K.tensor_amax(x)
    row = {"shape": [B, C, H, W]}
    outs = {}
    for sps in ("1", "3"):
        os.environ["CSEG_CONV3X3_SB_SPS"] = sps
        os.environ["CSEG_CONV3X3_SB16_CH"] = "none"          # force the main kernel on the 48-channel shape too
        row["sps" + sps + "_us"] = timeit(lambda: K.conv3x3_sb_run(x, w, False, None, 0, ax=ax))
        outs[sps] = K.conv3x3_sb_run(x, w, False, None, 0, ax=ax).clone()
    row["max_abs_diff"] = float((outs["1"] - outs["3"]).abs().max())
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), padding=1)
    row["err_vs_fp64"] = float((outs["3"][:1].double() - ref).abs().max())
    print(json.dumps(row), flush=True)
