"""Forward / backward-data time of the branch convolutions at the benched shapes under kernel-routing variants (environment
switches read per call by the library): which output-channel counts go to the 16-channel-chunk kernel, persistent form on / off.
One JSON line per (shape, variant). f16x3."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")
SHAPES = {"48": (8, 48, 128, 256), "64": (8, 64, 128, 256), "96": (8, 96, 64, 128), "192": (8, 192, 32, 64), "384": (8, 384, 16, 32)}
VARIANTS = {"default": {}, "one_tile": {"CSEG_CONV3X3_SB16_P": "0"}, "all_sb16p": {"CSEG_CONV3X3_SB16_CH": "48,96,192,384"},
            "all_sb16_one_tile": {"CSEG_CONV3X3_SB16_CH": "48,96,192,384", "CSEG_CONV3X3_SB16_P": "0"},
            "none_sb16": {"CSEG_CONV3X3_SB16_CH": "0"}}


def timeit(fn, iters=30, warm=5):
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


for name, (B, C, H, W) in SHAPES.items():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
    ref = torch.nn.functional.conv2d(x[:1], w, None, 1, 1)
    ax = K.tensor_amax(x)
    for vname, env in VARIANTS.items():
        for k in ("CSEG_CONV3X3_SB16_P", "CSEG_CONV3X3_SB16_CH"):
            os.environ.pop(k, None)
        os.environ.update(env)
        row = {"shape": name, "variant": vname}
        for nt in ((0,) if C not in K.CONV3X3_SB_PICK_NT_CHANNELS else (3, 6)):
            try:
                us = timeit(lambda: K.conv3x3_sb_run(x, w, False, None, nt, ax=ax))
                err = float((K.conv3x3_sb_run(x[:1].contiguous(), w, False, None, nt) - ref).abs().max())
                row["nt%d" % nt] = {"us": us, "err_vs_miopen": err}
            except Exception as e:      # noqa: BLE001
                row["nt%d" % nt] = {"error": repr(e)[:120]}
        print(json.dumps(row), flush=True)
