"""The 8 x 64-pixel head kernel (nt = CSEG_NT_SB8) against the 4 x 64-pixel kernel at the head's shape (720 -> 720, 8 x 128 x 256):
forward and backward-data, time per launch, deviation between the two and error against fp64 on one image."""
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
for (B, C, H, W) in ((8, 720, 128, 256), (1, 720, 128, 256), (8, 144, 64, 128)):
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
    row = {"shape": [B, C, H, W]}
    for name, nt in (("rows4", 0), ("rows8v1", K.NT_SB8), ("rows8", K.NT_SB8), ("rows8v1_again", K.NT_SB8), ("rows8_again", K.NT_SB8)):
        os.environ["CSEG_SB8_V"] = "1" if "v1" in name else "2"          # (csrc/conv3x3_sb16.hip reads the switch per call)
        row[name + "_fwd_us"] = timeit(lambda: K.conv3x3_sb_run(x, w, False, b, nt, ax=ax))
        row[name + "_bwd_us"] = timeit(lambda: K.conv3x3_sb_run(dy, w, True, None, nt, ax=ad))
    os.environ["CSEG_SB8_V"] = "1"
    y8v1, d8v1 = K.conv3x3_sb_run(x, w, False, b, K.NT_SB8, ax=ax), K.conv3x3_sb_run(dy, w, True, None, K.NT_SB8, ax=ad)
    os.environ["CSEG_SB8_V"] = "2"
    y4, y8 = K.conv3x3_sb_run(x, w, False, b, 0, ax=ax), K.conv3x3_sb_run(x, w, False, b, K.NT_SB8, ax=ax)
    row["v2_bit_identical_to_v1_fwd"] = bool(torch.equal(y8, y8v1))
    row["fwd_max_abs_diff_4_vs_8"] = float((y4 - y8).abs().max())
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), padding=1)
    row["rows8_err_vs_fp64"] = float((y8[:1].double() - ref).abs().max())
    row["rows4_err_vs_fp64"] = float((y4[:1].double() - ref).abs().max())
    d4, d8 = K.conv3x3_sb_run(dy, w, True, None, 0, ax=ad), K.conv3x3_sb_run(dy, w, True, None, K.NT_SB8, ax=ad)
    row["v2_bit_identical_to_v1_bwd"] = bool(torch.equal(d8, d8v1))
    row["bwd_rel_diff_4_vs_8"] = float((d4 - d8).abs().max() / d4.abs().max())
    print(json.dumps(row), flush=True)
