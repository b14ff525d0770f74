"""Debug aid: weight gradient of one small case against fp64, error per filter tap and per input-channel tile."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from contrastiveseg_amd import kernels as K

for case in [(1, 48, 48, 5, 64), (2, 16, 48, 9, 128), (1, 80, 96, 3, 64), (2, 96, 96, 16, 64)]:
    B, ci, co, H, W = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, ci, H, W, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    w64 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(dy.double())
    for rep in range(2):
        got = K.conv3x3_sb_wrw(x.cuda(), dy.cuda()).cpu().double()
        err = (got - w64.grad).abs()
        print(case, "rep", rep, "max err", float(err.max()), "per tap", [round(float(err[:, :, t // 3, t % 3].max()), 4) for t in range(9)],
              "per ci tile", [round(float(err[:, 16 * i:16 * i + 16].max()), 4) for i in range(ci // 16)], flush=True)
