"""1x1 convolutions of the projection head at the benched shape (8 x 128 x 256): forward, backward-data, weight gradient on the split
kernels, time per launch and deviation from torch's fp32 result."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
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
for (ci, co) in ((720, 720), (720, 256)):
    B, H, W = 8, 128, 256
    x = torch.randn(B, ci, H, W, generator=g).relu_().to(dev)
    w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(dev)
    dy = (torch.randn(B, co, H, W, generator=g) * 1e-3).to(dev)
    ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
    row = {"shape": [B, ci, co, H, W]}
    row["fwd_us"] = timeit(lambda: K.conv1x1_sb_run(x, w, False, None, ax=ax))
    row["bwd_us"] = timeit(lambda: K.conv1x1_sb_run(dy, w, True, None, ax=ad))
    row["wrw_us"] = timeit(lambda: K.conv1x1_sb_wrw(x, dy, ax=ax, ady=ad))
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    got = K.conv1x1_sb_wrw(x, dy, ax=ax, ady=ad)
    row["wrw_dev"] = float((got - ref).abs().max() / ref.abs().max())
    row["wrw_splits"] = K._hip.lib().cseg_conv1x1_sb_wrw_ws_floats(B, ci, co, H * W) // (ci * co)
    print(json.dumps(row), flush=True)
