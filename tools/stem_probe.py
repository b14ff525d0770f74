"""GPU: the first stem convolution (3 -> 64, stride 2) at the benched size, forward and weight gradient, on the fp32 kernels of
csrc/conv3x3_stem.hip against MIOpen (torch), HIP-event timings. Usage: python tools/stem_probe.py [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from contrastiveseg_amd import kernels as K  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    x = torch.randn(B, 3, 512, 1024, device="cuda")
    w = (torch.randn(64, 3, 3, 3, device="cuda") / 5).requires_grad_(True)
    dy = torch.randn(B, 64, 256, 512, device="cuda") * 1e-3
    y = K.conv3x3_s2_rgb(x, w)
    print("fwd  ours %.1f us   miopen %.1f us" % (timeit(lambda: K.conv3x3_s2_rgb(x, w)), timeit(lambda: F.conv2d(x, w, None, 2, 1))))

    def ours():
        w.grad = None
        y.backward(dy, retain_graph=True)
    yr = F.conv2d(x, w, None, 2, 1)

    def ref():
        w.grad = None
        yr.backward(dy, retain_graph=True)
    print("wrw  ours %.1f us   miopen %.1f us (backward of the one node, host side included)" % (timeit(ours), timeit(ref)))


if __name__ == "__main__":
    main()
