"""Times the 1x1 split-bf16 convolution (csrc/conv1x1_sb.hip) against rocBLAS / MIOpen fp32 at the benched shapes.
GPU box only."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import contrastiveseg_amd  # noqa: F401,E402
from contrastiveseg_amd import kernels as K  # noqa: E402

torch.backends.cudnn.benchmark = False
SHAPES = [("proj_720_720", 8, 720, 720, 128, 256), ("proj_720_256", 8, 720, 256, 128, 256),
          ("layer1_64_256", 8, 64, 256, 128, 256), ("layer1_256_64", 8, 256, 64, 128, 256)]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    only = sys.argv[1:] or [s[0] for s in SHAPES]
    for name, B, ci, co, H, W in SHAPES:
        if name not in only:
            continue
        x = torch.randn(B, ci, H, W, device="cuda")
        dy = torch.randn(B, co, H, W, device="cuda")
        w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
        flops = 2.0 * B * H * W * ci * co
        rows = [("split_bf16 fwd (pack + conv)", timeit(lambda: K.conv1x1_sb_run(x, w, False))),
                ("split_bf16 bwd_data (pack + conv)", timeit(lambda: K.conv1x1_sb_run(dy, w, True))),
                ("split_bf16 bwd_weight (kernel + reduction)", timeit(lambda: K.conv1x1_sb_wrw(x, dy))),
                ("fp32 fwd (torch)", timeit(lambda: F.conv2d(x, w))),
                ("fp32 bwd_data (torch)", timeit(lambda: torch.ops.aten.convolution_backward(
                    dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])),
                ("fp32 bwd_weight (torch)", timeit(lambda: torch.ops.aten.convolution_backward(
                    dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]))]
        ref = F.conv2d(x[:1].double(), w.double())
        e_sb = float((K.conv1x1_sb_run(x[:1].contiguous(), w, False).double() - ref).abs().max())
        e_32 = float((F.conv2d(x[:1], w).double() - ref).abs().max())
        for tag, us in rows:
            print(json.dumps({"shape": name, "dims": [B, ci, co, H, W], "kernel": tag, "us": round(us, 1),
                              "fp32_equiv_TFLOPs": round(flops / us / 1e6, 1)}), flush=True)
        print(json.dumps({"shape": name, "max_abs_err_vs_fp64": {"split_bf16": e_sb, "fp32": e_32},
                          "out_absmax": float(ref.abs().max())}), flush=True)


if __name__ == "__main__":
    main()
