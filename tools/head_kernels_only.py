"""The three dominant kernels of the benched step in isolation (720 -> 720 3x3 at 8x128x256: forward, backward-data, weight
gradient; current arithmetic), a few launches each: the target of `rocprofv3 --pmc ... -- python tools/head_kernels_only.py`.
Optional args: branch  -> also the 48 / 96 / 192-channel branch shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from contrastiveseg_amd import kernels as K

dev = torch.device("cuda:0")
shapes = [(8, 720, 128, 256)]
if "branch" in sys.argv[1:]:
    shapes += [(8, 48, 128, 256), (8, 96, 64, 128), (8, 192, 32, 64), (8, 384, 16, 32)]
for B, C, H, W in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dev)
    nt = K.conv3x3_sb_pick_nt(x, C) if C in K.CONV3X3_SB_PICK_NT_CHANNELS else 0
    ax, ad = (K.tensor_amax(x), K.tensor_amax(dy)) if K.split_arith_id() else (None, None)
    for _ in range(3):
        K.conv3x3_sb_run(x, w, False, None, nt, ax=ax)
        K.conv3x3_sb_run(dy, w, True, None, nt, ax=ad)
        K.conv3x3_sb_wrw(x, dy, ax=ax, ady=ad)
    torch.cuda.synchronize()
print("done")
