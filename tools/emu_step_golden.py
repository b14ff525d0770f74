"""One-SGD-step golden (tests/test_step_golden.py: reference fp64 truth + the reference's own fp32 deviation) with the
device = the CPU emulation of the execution model (tests/emu) and EVERY split-bf16 kernel forced on regardless of the
grid-filling thresholds: 3x3 forward / backward-data on the 48 / 96-channel branches and the 720-channel head, the
split-bf16 weight gradient (version per CSEG_CONV3X3_SB_WRW_V), the 1x1 forward / backward-data / weight-gradient
kernels, optionally the row-sparse projection-head backward. The step goldens' own shapes (B=2, 128x256 input) stay below
those thresholds on the GPU, so this is the only place where the whole-network backward runs through all of them before
hardware time is spent on it. Minutes per case (every MFMA is a fiber rendezvous).

    python tools/emu_step_golden.py [case] [--wrw 0|1|2] [--c1 0|1] [--c1wrw 0|1] [--sparse 0|1] [--channels 48,96]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Patch(object):
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", nargs="?", default="step_hrnet48_contrast")
    ap.add_argument("--wrw", type=int, default=1)
    ap.add_argument("--c1", type=int, default=1)
    ap.add_argument("--c1wrw", type=int, default=1)
    ap.add_argument("--sparse", type=int, default=0)
    ap.add_argument("--channels", default="48,96")
    a = ap.parse_args()
    if a.wrw:
        os.environ["CSEG_CONV3X3_SB_WRW_V"] = str(a.wrw)
    import numpy as np
    import torch
    from tests.emu import inject
    inject.install(_Patch())
    from contrastiveseg_amd import kernels as K
    K.CONV3X3_SPLIT_BF16 = True
    K.CONV3X3_SB_MIN_TILES = 1
    K.CONV3X3_SB_BRANCH_CHANNELS = tuple(int(c) for c in a.channels.split(","))
    K.CONV3X3_SB_WRW = bool(a.wrw)
    K.CONV1X1_SPLIT_BF16 = bool(a.c1)
    K.CONV1X1_SB_MIN_TILES = 1
    K.CONV1X1_SB_WRW = bool(a.c1wrw)
    K.SPARSE_EMBED_GRAD = bool(a.sparse)
    calls = {}
    for name in ("conv3x3_sb_run", "conv3x3_sb_wrw", "conv1x1_sb_run", "conv1x1_sb_wrw", "_conv3x3_wrw"):
        def wrap(fn, name=name):
            def f(*args, **kw):
                calls[name] = calls.get(name, 0) + 1
                return fn(*args, **kw)
            return f
        setattr(K, name, wrap(getattr(K, name)))
    import test_step_golden as T
    from oracle.make_golden import STEP_CASES
    c = STEP_CASES[a.case]
    g = np.load(os.path.join(ROOT, "tests", "golden", "%s.npz" % a.case))
    t0 = time.time()
    res = T._run(c, torch.device("cpu"))
    worst = T._compare(res, g, c, 1e-3, 1e-3, 5e-2)
    out = {"case": a.case, "switches": vars(a), "kernel_calls": calls, "seconds": round(time.time() - t0, 1),
           "loss0": res["loss0"], "loss0_ref": float(g["loss0"]), "loss1": res["loss1"], "loss1_ref": float(g["loss1"]),
           "grad_rel_l2_vs_fp64 (bound)": {k: "%.2e (%.2e)" % v for k, v in worst.items()},
           "times_reference_noise": {k: round(v[0] / max(float(g["gradnoise_l2/" + k]), 1e-30), 2) for k, v in worst.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
