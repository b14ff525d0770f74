"""Device time of the contrastive forward, three launches (s_gemm + row_pass + mean, S through HBM) vs the fused single launch, at the
benched shapes (self: N = 912 anchors, D = 256; bank: 1024 x 4104). Run under `rocprofv3 --kernel-trace --stats` for per-kernel
durations (tools/gpu_job.sh step `contrast`); prints HIP-event times of back-to-back calls as well (host-bound for the short ones)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contrastiveseg_amd import kernels as K      # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
D, Kc = 256, 19


def unit(*shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=g), dim=-1).to(dev)


cases = {}
A = unit(912, D)
lab = torch.randint(0, Kc, (152,), generator=g).repeat(6).int().to(dev)        # view-major rows: label of row r = y[r % T]
cases["self N=912"] = K._desc(0, A, lab, 0.1, 0.07)
A2 = unit(1024, D)
lab2 = torch.randint(0, Kc, (1024,), generator=g).int().to(dev)
sq, pq = unit(Kc, 108, D), unit(Kc, 108, D)
cases["bank 1024x4104"] = K._desc(2, A2, lab2, 0.1, 0.07, None, None, sq, pq)
sq5, pq5 = unit(Kc, 5000, D), unit(Kc, 5000, D)
A3 = unit(152, D)
lab3 = torch.randint(0, Kc, (152,), generator=g).int().to(dev)
cases["bank 152x190000 (reference config)"] = K._desc(2, A3, lab3, 0.07, 0.07, None, None, sq5, pq5)
out = {}
for name, d in cases.items():
    res = {}
    for tag, flag in (("three", "0"), ("fused", "1")):
        K.CONTRAST_FUSED = flag
        for _ in range(5):
            loss, _ = K.contrast_forward(d, dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            loss, _ = K.contrast_forward(d, dev)
        e1.record()
        torch.cuda.synchronize()
        res[tag] = {"us_per_call_host_and_device": round(e0.elapsed_time(e1) * 1e3 / 50, 1), "loss": float(loss)}
    res["rel_diff"] = abs(res["three"]["loss"] - res["fused"]["loss"]) / abs(res["three"]["loss"])
    out[name] = res
print(json.dumps(out))
