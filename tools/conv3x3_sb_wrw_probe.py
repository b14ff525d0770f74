"""Times the split-bf16 weight-gradient kernel (csrc/conv3x3_sb_wrw.hip) against MIOpen's fp32 weight gradient and the
fp32-MFMA kernel at the benched shapes (one JSON line per measurement). GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import contrastiveseg_amd  # noqa: F401,E402
from contrastiveseg_amd import kernels as K  # noqa: E402

torch.backends.cudnn.benchmark = False
SHAPES = [("branch_48", 8, 48, 128, 256), ("branch_96", 8, 96, 64, 128), ("branch_192", 8, 192, 32, 64),
          ("head_720", 8, 720, 128, 256)]


def timeit(fn, iters):
    for _ in range(2):
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
    for name, B, C, H, W in SHAPES:
        if name not in only:
            continue
        x = torch.randn(B, C, H, W, device="cuda")
        dy = torch.randn(B, C, H, W, device="cuda")
        w = torch.zeros(C, C, 3, 3, device="cuda")
        flops = 2.0 * B * H * W * C * C * 9
        iters = 3 if C >= 720 else 10

        def miopen():
            return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        rows = []
        for v in ("1", "2"):
            os.environ["CSEG_CONV3X3_SB_WRW_V"] = v
            rows.append(("split_bf16 wrw v%s" % v, timeit(lambda: K.conv3x3_sb_wrw(x, dy), iters)))
        rows.append(("miopen fp32 wrw", timeit(miopen, iters)))
        if C in (48, 96):
            rows.append(("fp32-MFMA wrw", timeit(lambda: K._conv3x3_wrw(x, dy, C, C), iters)))
        ref = miopen()
        got = K.conv3x3_sb_wrw(x, dy)
        for tag, us in rows:
            print(json.dumps({"shape": name, "dims": [B, C, H, W], "kernel": tag, "us": round(us, 1),
                              "fp32_equiv_TFLOPs": round(flops / us / 1e6, 1)}), flush=True)
        print(json.dumps({"shape": name, "max_abs_diff_vs_miopen_fp32": float((got - ref).abs().max()),
                          "grad_absmax": float(ref.abs().max())}), flush=True)


if __name__ == "__main__":
    main()
