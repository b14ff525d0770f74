"""Where does the time of the three head kernels go? Ablation builds (CSEG_ABLATE bits, wrong results, timing only) of the f16x3
3x3 forward kernel (NT = 9) and of the weight gradient at 720 channels, 8x128x256:
  forward: 1 = no MFMAs, 2 = no split arithmetic, 4 = no patch loads, 8 = no weight DMA;  weight gradient: 1, 2, 4 likewise.
One JSON line per kernel: {variant: us}."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")
B, C, H, W = 8, 720, 128, 256


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
x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
row = {"kernel": "conv3x3 720->720 forward, f16x3"}
for abl in (0, 14):
    os.environ["CSEG_ABLATE"] = str(abl)
    row[str(abl)] = timeit(lambda: K.conv3x3_sb_run(x, w, False, None, 0, ax=ax))
print(json.dumps(row), flush=True)
row = {"kernel": "conv3x3 720->720 weight gradient, f16x3"}
for abl in (0, 2, 4, 6, 14):
    os.environ["CSEG_ABLATE"] = str(abl)
    row[str(abl)] = timeit(lambda: K.conv3x3_sb_wrw(x, dy, ax=ax, ady=ad))
print(json.dumps(row), flush=True)
os.environ.pop("CSEG_ABLATE")
