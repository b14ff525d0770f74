"""CPU study for the next round's main lever: how far are the HRNet-W48-contrast logits from the
fp32 reference when every convolution is computed from bf16-rounded operands with fp32 accumulation, as the bf16 MFMA
would, using 1, 2 or 3 split terms per product?

  x = x_hi + x_lo,  x_hi = bf16(x), x_lo = bf16(x - x_hi)   (same for w)
  1 term : x_hi*w_hi                      (plain bf16 inputs)
  2 terms: + x_lo*w_hi                    (activation residual only)
  3 terms: + x_hi*w_lo                    (drops only x_lo*w_lo ~ 2^-16 relative)
  4 terms: + x_lo*w_lo;  6 terms: three-way split, all products down to ~2^-24

Runs the repo's model classes on CPU (device half = oracle/cpu_port.py) on the seeded golden input and prints
max |logit - fp32 logit| / max |fp32 logit| -- to be compared with the 1e-3 bar of BASELINE.json."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_port  # noqa: E402
from oracle.make_golden import MODEL_CASES, freeze_dropout, model_input  # noqa: E402

_real_conv = F.conv2d
TERMS = 0


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split_conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if TERMS == 0:
        return _real_conv(x, w, b, stride, padding, dilation, groups)
    xh, wh = _bf(x), _bf(w)
    y = _real_conv(xh, wh, None, stride, padding, dilation, groups)
    if TERMS >= 2:
        y = y + _real_conv(_bf(x - xh), wh, None, stride, padding, dilation, groups)
    if TERMS >= 3:
        y = y + _real_conv(xh, _bf(w - wh), None, stride, padding, dilation, groups)
    if TERMS >= 4:
        y = y + _real_conv(_bf(x - xh), _bf(w - wh), None, stride, padding, dilation, groups)
    if TERMS >= 6:     # three-way split x = x1 + x2 + x3: add the two products that involve the third parts
        x3 = _bf(x - xh - _bf(x - xh))
        w3 = _bf(w - wh - _bf(w - wh))
        y = y + _real_conv(x3, wh, None, stride, padding, dilation, groups) \
              + _real_conv(xh, w3, None, stride, padding, dilation, groups)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def main():
    global TERMS
    cpu_port.install(None)
    from contrastiveseg_amd.lib.models.model_manager import ModelManager
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    name = "hrnet_w48_contrast"
    c = MODEL_CASES[name]
    cfg = Configer(config_dict={"data": {"num_classes": c["K"]},
                                "network": {"backbone": c["backbone"], "model_name": name, "bn_type": "torchsyncbn",
                                            "resume": None, "pretrained": None, "multi_grid": [1, 1, 1]},
                                "contrast": {"proj_dim": 256}})
    torch.manual_seed(304)
    net = ModelManager(cfg).semantic_segmentor().train()
    freeze_dropout(net)
    x = torch.from_numpy(model_input(c))
    torch.nn.functional.conv2d = split_conv
    import torch.nn.modules.conv as convmod
    convmod.F.conv2d = split_conv
    out = {}
    with torch.no_grad():
        for t in (0, 1, 2, 3, 4, 6):
            TERMS = t
            o = net(x, with_embed=True)
            out[t] = (o["seg"].clone(), o["embed"].clone())
    ref_seg, ref_emb = out[0]
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % name))
    print("fp32 vs reference golden: %.2e" % float((ref_seg - torch.from_numpy(g["seg"])).abs().max()))
    for t in (1, 2, 3, 4, 6):
        seg, emb = out[t]
        print("terms=%d  logits: max abs err %.3e  (rel to max |logit| %.3e)   embed: max abs err %.3e" % (
            t, float((seg - ref_seg).abs().max()), float((seg - ref_seg).abs().max() / ref_seg.abs().max()),
            float((emb - ref_emb).abs().max())))


if __name__ == "__main__":
    main()
