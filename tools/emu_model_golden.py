"""Model-forward goldens (tests/test_models_golden.py: logits produced by the reference itself) with the device = the CPU
emulation of the execution model (tests/emu) and the split-bf16 convolutions engaged regardless of the grid-fill thresholds:
the north_star bar "pixel logits within 1e-3" checked with the HIP SOURCES of the head / branch convolutions, fused BN,
upsample+concat and exchange fusion doing the device work. One JSON line per case. Minutes per case.

    python tools/emu_model_golden.py [case ...]
"""
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
    import numpy as np
    from tests.emu import inject
    inject.install(_Patch())
    from contrastiveseg_amd import kernels as K
    K.CONV3X3_SB_MIN_TILES = 1
    calls = {}
    for name in ("conv3x3_sb_run", "bn_fwd", "upsample_concat", "fuse_sum_relu"):
        def wrap(fn, name=name):
            def f(*a, **k):
                calls[name] = calls.get(name, 0) + 1
                return fn(*a, **k)
            return f
        setattr(K, name, wrap(getattr(K, name)))
    import contrastiveseg_amd.lib.models.nets.hrnet as nh
    import contrastiveseg_amd.lib.models.backbones.hrnet_backbone as hb
    import contrastiveseg_amd.lib.models.tools.fused_bn as fb
    for m in (nh, hb, fb):                      # the modules hold their own reference to the wrappers' module: same object
        assert m.K is K
    import test_models_golden as T
    from oracle.make_golden import MODEL_CASES
    for case in (sys.argv[1:] or T.CPU_CASES):
        c = MODEL_CASES[case]
        g = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % case))
        calls.clear()
        t0 = time.time()
        out = T._forward(T._build(case, c), c, "cpu")
        T._check(out, g, 1e-3)
        seg = out["seg"].detach().numpy()
        print(json.dumps({"case": case, "max_abs_logit_error": float(np.abs(seg - g["seg"]).max()),
                          "logit_absmax": float(np.abs(g["seg"]).max()), "reference_fp32_vs_fp64": float(g["seg_fp32_noise"]),
                          "bound": 1e-3, "kernel_calls": dict(calls), "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
