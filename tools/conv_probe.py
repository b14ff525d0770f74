"""Times one convolution (forward, backward-data, backward-weight) through PyTorch/MIOpen on the current GPU.
Used to choose MIOpen solver settings for the layers that dominate the HRNet-W48 step (rocprof: profiles/)."""
import argparse
import json
import os
import time

import torch
import torch.nn.functional as F


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--shape", default="8,720,128,256")       # N,C,H,W
    p.add_argument("--out-ch", type=int, default=720)
    p.add_argument("--k", type=int, default=3)
    p.add_argument("--stride", type=int, default=1)
    p.add_argument("--dil", type=int, default=1)
    p.add_argument("--find", type=int, default=0)
    p.add_argument("--iters", type=int, default=5)
    p.add_argument("--tag", default="")
    p.add_argument("--immediate", type=int, default=0)
    a = p.parse_args()
    torch.backends.cudnn.benchmark = bool(a.find)
    if a.immediate:
        torch.backends.miopen.immediate = True
    N, C, H, W = [int(v) for v in a.shape.split(",")]
    dev = "cuda"
    x = torch.randn(N, C, H, W, device=dev, requires_grad=True)
    w = torch.randn(a.out_ch, C, a.k, a.k, device=dev, requires_grad=True) * 0.01
    w = w.detach().requires_grad_(True)
    pad = a.dil * (a.k - 1) // 2

    def fwd():
        return F.conv2d(x, w, None, a.stride, pad, a.dil)

    def ev(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    t0 = time.time()
    y = fwd()
    g = torch.randn_like(y)
    torch.autograd.grad(y, (x, w), g)
    torch.cuda.synchronize()
    t_first = time.time() - t0
    y = fwd()
    ms_f = ev(fwd, a.iters)
    ms_bd = ev(lambda: torch.autograd.grad(y, x, g, retain_graph=True), a.iters)
    ms_bw = ev(lambda: torch.autograd.grad(y, w, g, retain_graph=True), a.iters)
    flops = 2.0 * N * a.out_ch * C * a.k * a.k * (H // a.stride) * (W // a.stride)
    print(json.dumps({"tag": a.tag, "shape": a.shape, "out": a.out_ch, "k": a.k, "find": a.find,
                      "first_call_s": round(t_first, 1), "fwd_ms": round(ms_f, 2), "bwd_data_ms": round(ms_bd, 2),
                      "bwd_weight_ms": round(ms_bw, 2), "fwd_TF": round(flops / ms_f * 1e-9, 1),
                      "bwd_data_TF": round(flops / ms_bd * 1e-9, 1), "bwd_weight_TF": round(flops / ms_bw * 1e-9, 1),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")}}))


if __name__ == "__main__":
    main()
