"""Numerical probe (build container only: needs /root/reference): how far do the logits of the reference HRNet-W48
contrast model move when every convolution is evaluated with split-bf16 operands (the arithmetic a bf16-MFMA
emulation of fp32 would perform: products of bf16 pieces are exact in fp32, accumulation in fp32) instead of fp32?
Ground truth = the same network in fp64. Variants: fp32, bf16x1 (plain bf16 operands), bf16x3 (hi*hi + hi*mid + mid*hi),
bf16x6 (all terms down to 2^-24), tf32 (operands rounded to 10 mantissa bits: what the reference's convolutions use by
default on Ampere-class GPUs). Writes one JSON line per variant. Round-3 input: is a bf16 split MFMA path inside the
north_star bar (logits within 1e-3)?"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import ref_shim  # noqa: E402


def split3(t):
    hi = t.bfloat16().float()
    r = t - hi
    mid = r.bfloat16().float()
    lo = (r - mid).bfloat16().float()
    return hi, mid, lo


def split_h2(t):
    """two fp16 pieces of t * 2^k (k puts max|t| into [2^14, 2^15)): hi = rn16(t s), lo = rn16(t s - hi); -> hi, lo, 1/s"""
    amax = float(t.abs().max())
    k = 14 - int(np.floor(np.log2(amax))) if amax > 0 else 0
    s = 2.0 ** k
    ts = t * s
    hi = ts.half().float()
    lo = (ts - hi).half().float()
    return hi, lo, 1.0 / s


def tf32(t):
    i = t.contiguous().view(torch.int32)
    i = (i + 0x1000) & ~0x1FFF          # round to nearest at 10 explicit mantissa bits
    return i.view(torch.float32)


MODE = {"v": "fp32"}
_orig = torch.nn.Conv2d._conv_forward


def conv_forward(self, x, w, b):
    mode = MODE["v"]
    if mode == "fp32" or x.dtype != torch.float32:
        return _orig(self, x, w, b)

    def c(a, ww, bias=None):
        return F.conv2d(a, ww, bias, self.stride, self.padding, self.dilation, self.groups)
    if mode == "tf32":
        return c(tf32(x), tf32(w), b)
    if mode in ("fp16x3", "fp16x4"):
        xh, xl, xi = split_h2(x)
        wh, wl, wi = split_h2(w)
        out = c(xl, wh) + c(xh, wl)
        if mode == "fp16x4":
            out = out + c(xl, wl)
        out = (out + c(xh, wh)) * (xi * wi)
        return out if b is None else out + b.view(1, -1, 1, 1)
    xh, xm, xl = split3(x)
    wh, wm, wl = split3(w)
    if mode == "bf16x1":
        return c(xh, wh, b)
    out = c(xm, wh) + c(xh, wm)
    if mode == "bf16x6":
        out = out + (c(xl, wh) + c(xm, wm) + c(xh, wl))
    return out + c(xh, wh, b)            # smallest terms first


def main():
    H, W, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 256, 2)))
    ref_shim.install()
    from lib.models.model_manager import ModelManager
    cfg = ref_shim.configer(num_classes=19, model_name="hrnet_w48_contrast", backbone="hrnet48", contrast={})
    torch.manual_seed(304)
    net = ModelManager(cfg).semantic_segmentor().train()
    for m in net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    x = torch.from_numpy(np.random.RandomState(34).standard_normal((B, 3, H, W)).astype(np.float32))
    torch.nn.Conv2d._conv_forward = conv_forward
    outs = {}
    with torch.no_grad():
        for mode in ("fp32", "tf32", "bf16x1", "bf16x3", "bf16x6", "fp16x3", "fp16x4"):
            MODE["v"] = mode
            o = net(x, with_embed=True)
            outs[mode] = (o["seg"].double(), o["embed"].double())
        MODE["v"] = "fp32"
        net64 = net.double()
        o = net64(x.double(), with_embed=True)
        truth = (o["seg"], o["embed"])
    for mode, (seg, emb) in outs.items():
        print(json.dumps({"variant": mode, "input": [B, 3, H, W], "logit_absmax": round(float(truth[0].abs().max()), 4),
                          "seg_max_abs_err_vs_fp64": float((seg - truth[0]).abs().max()),
                          "seg_rms_err": float((seg - truth[0]).pow(2).mean().sqrt()),
                          "embed_max_abs_err_vs_fp64": float((emb - truth[1]).abs().max())}), flush=True)


if __name__ == "__main__":
    main()
