"""Times the split-bf16 3x3 convolution (csrc/conv3x3_sb.hip) against MIOpen's fp32 path and the fp32-MFMA kernel at the
benched shapes (one JSON line per measurement). GPU box only."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import contrastiveseg_amd  # noqa: F401,E402  (points MIOpen at the shipped solver records)
from contrastiveseg_amd import kernels as K  # noqa: E402

torch.backends.cudnn.benchmark = False
SHAPES = [  # name, B, C, H, W
    ("head_720", 8, 720, 128, 256),
    ("branch_48", 8, 48, 128, 256),
    ("branch_96", 8, 96, 64, 128),
    ("branch_192", 8, 192, 32, 64),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    only = sys.argv[1:] or [s[0] for s in SHAPES]
    for name, B, C, H, W in SHAPES:
        if name not in only:
            continue
        x = torch.randn(B, C, H, W, device="cuda")
        w = torch.randn(C, C, 3, 3, device="cuda") / (3.0 * C ** 0.5)
        flops = 2.0 * B * H * W * C * C * 9
        iters = 5 if C >= 720 else 20
        rows = []
        for glds in ("1", "0"):
            os.environ["CSEG_CONV3X3_SB_GLDS"] = glds
            for flip, tag in ((False, "fwd"), (True, "bwd_data")):
                us = timeit(lambda: K.conv3x3_sb_run(x, w, flip), iters)
                rows.append(("split_bf16 glds=%s %s (incl. weight packing)" % (glds, tag), us))
        os.environ["CSEG_CONV3X3_SB_GLDS"] = "1"
        e_var = {}
        # variant 1: buffer-load addressing of the patch (no spills at NT = 9); variant 2: 16-channel chunks + two blocks per
        # CU for <= 192 channels (csrc/conv3x3_sb16.hip), variant 1 above that
        for var in ("1", "2"):
            if var == "2" and C > 192:
                continue
            os.environ["CSEG_CONV3X3_SB_VAR"] = var
            for flip, tag in ((False, "fwd"), (True, "bwd_data")):
                rows.append(("split_bf16 var=%s %s (incl. weight packing)" % (var, tag),
                             timeit(lambda: K.conv3x3_sb_run(x, w, flip), iters)))
            e_var[var] = float((K.conv3x3_sb_run(x[:1].contiguous(), w, False).double()
                                - F.conv2d(x[:1].double(), w.double(), None, 1, 1)).abs().max())
        del os.environ["CSEG_CONV3X3_SB_VAR"]
        e_v1 = e_var["1"]
        rows.append(("miopen fp32 fwd", timeit(lambda: F.conv2d(x, w, None, 1, 1), iters)))
        if C in (48, 96, 192):
            rows.append(("fp32-MFMA kernel fwd (incl. weight packing)", timeit(lambda: K._conv3x3_run(x, w, False), iters)))
        ref = F.conv2d(x[:1].double(), w.double(), None, 1, 1)
        e_sb = float((K.conv3x3_sb_run(x[:1].contiguous(), w, False).double() - ref).abs().max())
        e_32 = float((F.conv2d(x[:1], w, None, 1, 1).double() - ref).abs().max())
        for tag, us in rows:
            print(json.dumps({"shape": name, "dims": [B, C, H, W], "kernel": tag, "us": round(us, 1),
                              "fp32_equiv_TFLOPs": round(flops / us / 1e6, 1)}), flush=True)
        print(json.dumps({"shape": name, "max_abs_err_vs_fp64": {"split_bf16": e_sb, "split_bf16_var1": e_v1, "split_bf16_var2": e_var.get("2"), "miopen_fp32": e_32},
                          "out_absmax": float(ref.abs().max())}), flush=True)


if __name__ == "__main__":
    main()
