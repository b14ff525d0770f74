"""3x3 / stride 2 / pad 1 at HRNet-W48's benched shapes (batch 8 at 1024 x 512): the split f16x3 kernels (forward, backward-data,
weight gradient) against MIOpen through aten, per direction, plus the deviation between the two. One JSON line per shape."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch
import torch.nn.functional as F

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
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


SHAPES = [(48, 96, 64, 128), (48, 48, 64, 128), (48, 192, 32, 64), (96, 192, 32, 64), (96, 96, 32, 64), (48, 48, 32, 64),
          (48, 384, 16, 32), (96, 384, 16, 32), (192, 384, 16, 32), (256, 96, 64, 128)]
g = torch.Generator().manual_seed(1)
for (ci, co, Ho, Wo) in SHAPES:
    B = 8
    x = torch.randn(B, ci, 2 * Ho, 2 * Wo, generator=g).relu_().to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(dev)
    dy = (torch.randn(B, co, Ho, Wo, generator=g) * 1e-3).to(dev)
    ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
    row = {"shape": [B, ci, co, Ho, Wo]}
    bw = lambda mask: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, mask)
    row["fwd_us"] = [timeit(lambda: K.conv3x3_s2_run(x, w, ax=ax)), timeit(lambda: F.conv2d(x, w, None, 2, 1))]
    row["fwd_dev"] = float((K.conv3x3_s2_run(x, w, ax=ax) - F.conv2d(x, w, None, 2, 1)).abs().max() / F.conv2d(x, w, None, 2, 1).abs().max())
    if ci % 48 == 0:
        row["bwd_us"] = [timeit(lambda: K.conv3x3_s2_bwd_run(dy, w, ady=ad)), timeit(lambda: bw([True, False, False]))]
        ref = bw([True, False, False])[0]
        row["bwd_dev"] = float((K.conv3x3_s2_bwd_run(dy, w, ady=ad) - ref).abs().max() / ref.abs().max())
    row["wrw_us"] = [timeit(lambda: K.conv3x3_s2_wrw(x, dy, ax=ax, ady=ad)), timeit(lambda: bw([False, True, False]))]
    ref = bw([False, True, False])[1]
    row["wrw_dev"] = float((K.conv3x3_s2_wrw(x, dy, ax=ax, ady=ad) - ref).abs().max() / ref.abs().max())
    print(json.dumps(row), flush=True)
