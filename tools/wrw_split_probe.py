"""Number of pixel splits of the 3x3 weight gradient (CSEG_SB_WRW_SPLITS forces it; default = the library's cost model, conv3x3_sb_wrw.hip:
sb_wrw_splits): time of kernel + reduction at the benched shapes for a few values around the model's choice. One JSON line per shape."""
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
CASES = [((8, 720, 128, 256), (0, 3, 4, 5, 8, 16)), ((8, 384, 16, 32), (0, 2, 4, 8, 16)), ((8, 192, 32, 64), (0, 4, 8, 16)),
         ((8, 96, 64, 128), (0, 16, 32, 64)), ((8, 48, 128, 256), (0, 64, 128, 256))]
MODEL_ONLY = os.environ.get("CSEG_PROBE_MODEL_ONLY") == "1"
for (B, C, H, W), ns in CASES:
    ns = (0,) if MODEL_ONLY else ns
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
    ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
    row = {"shape": [B, C, H, W]}
    for n in ns:
        if n:
            os.environ["CSEG_SB_WRW_SPLITS"] = str(n)
        else:
            os.environ.pop("CSEG_SB_WRW_SPLITS", None)
        row["model" if n == 0 else str(n)] = timeit(lambda: K.conv3x3_sb_wrw(x, dy, ax=ax, ady=ad))
    os.environ.pop("CSEG_SB_WRW_SPLITS", None)
    row["model_splits"] = K._hip.lib().cseg_conv3x3_sb_wrw_ws_floats(B, C, C, H, W) // (9 * C * C)
    print(json.dumps(row), flush=True)
