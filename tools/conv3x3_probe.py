"""Correctness + timing of the hand-written 3x3 MFMA convolution (csrc/conv3x3.hip) against MIOpen through PyTorch,
forward / backward-data (and backward-weight when available), at the HRNet-W48 branch shapes of the benched step."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    import contrastiveseg_amd  # noqa: F401
    from contrastiveseg_amd import kernels as K
    torch.backends.cudnn.benchmark = False
    dev = "cuda"
    # small exactness check vs fp64 CPU
    g = torch.Generator().manual_seed(0)
    for (B, C, Co, H, W) in [(2, 48, 48, 9, 12), (1, 96, 48, 7, 68), (2, 48, 96, 5, 132)]:
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(Co, C, 3, 3, generator=g) * 0.1
        gy = torch.randn(B, Co, H, W, generator=g)
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, 1, 1)
        yr.backward(gy.double())
        xd = x.to(dev).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        y = K.conv3x3(xd, wd)
        y.backward(gy.to(dev))
        print(json.dumps({"check": [B, C, Co, H, W], "fwd_err": float((y.detach().cpu().double() - yr.detach()).abs().max()),
                          "dx_err": float((xd.grad.cpu().double() - xr.grad).abs().max()),
                          "dw_err": float((wd.grad.cpu().double() - wr.grad).abs().max()),
                          "scale": float(yr.abs().max())}), flush=True)
    for (B, C, H, W) in [(8, 48, 128, 256), (8, 96, 64, 128), (1, 48, 128, 256)]:
        x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
        w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).requires_grad_(True)
        y_m = F.conv2d(x, w, None, 1, 1)
        gy = torch.randn_like(y_m)
        torch.autograd.grad(y_m, (x, w), gy, retain_graph=True)
        y_k = K.conv3x3(x, w)
        flops = 2.0 * B * C * C * 9 * H * W
        r = {"shape": [B, C, H, W], "max_abs_diff_vs_miopen": float((y_k - y_m).abs().max()),
             "miopen_fwd_us": round(ev(lambda: F.conv2d(x, w, None, 1, 1)), 1),
             "cseg_fwd_us": round(ev(lambda: K.conv3x3(x, w)), 1),
             "miopen_bwd_data_us": round(ev(lambda: torch.autograd.grad(y_m, x, gy, retain_graph=True)), 1),
             "cseg_bwd_data_us": round(ev(lambda: torch.autograd.grad(y_k, x, gy, retain_graph=True)), 1),
             "miopen_bwd_weight_us": round(ev(lambda: torch.autograd.grad(y_m, w, gy, retain_graph=True)), 1),
             "cseg_bwd_weight_us": round(ev(lambda: K._conv3x3_wrw(x.detach(), gy, C, C)), 1)}
        dw_m = torch.autograd.grad(y_m, w, gy, retain_graph=True)[0]
        dw_k = K._conv3x3_wrw(x.detach(), gy, C, C)
        r["dw_rel_diff_vs_miopen"] = float((dw_k - dw_m).abs().max() / dw_m.abs().max())
        # pure kernel times of the two data-path kernels (autograd.grad above also runs the weight gradient)
        r["cseg_fwd_kernel_us"] = round(ev(lambda: K._conv3x3_run(x.detach(), w.detach(), False)), 1)
        r["cseg_bwd_data_kernel_us"] = round(ev(lambda: K._conv3x3_run(gy, w.detach(), True)), 1)
        r["cseg_fwd_TF"] = round(flops / r["cseg_fwd_us"] * 1e-6, 1)
        r["miopen_fwd_TF"] = round(flops / r["miopen_fwd_us"] * 1e-6, 1)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
