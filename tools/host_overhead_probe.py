"""Host cost of the autograd / ctypes path with the kernels stubbed out (CPU, no GPU needed): the library's entry points become
no-ops, tensors are tiny, so what remains is Python + autograd + allocator time per call -- the part of the step that limits a
batch-8 step once the GPU work overlaps (tools/host_profile.py on the MI355X: 79 ms of enqueue time per step).
  python tools/host_overhead_probe.py            us per residual block (forward + backward), fused node vs four nodes, + cProfile top"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


class _Patch(object):
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    from tests.emu import inject
    from contrastiveseg_amd import _hip
    from contrastiveseg_amd import kernels as K
    inject.install(_Patch())
    _hip.call = lambda name, *a: None                      # no device work at all
    K.CONV3X3_SB_MIN_TILES = 1
    from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
    from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
    torch.manual_seed(0)
    blocks = [mark_conv_bn_pairs(BasicBlock(48, 48, bn_type="torchbn").train()) for _ in range(4)]
    x0 = torch.randn(2, 48, 4, 64)
    gy = torch.randn(2, 48, 4, 64)

    def step():
        x = (x0 * 1.0).requires_grad_(True)
        y = x * 1.0
        for b in blocks:
            y = b(y)
        y.backward(gy)

    for fused in (True, False):
        K.BLOCK_FUSED = fused
        for _ in range(20):
            step()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        dt = (time.perf_counter() - t0) / n / len(blocks) * 1e6
        print("%s: %.1f us per residual block (forward + backward)" % ("one node " if fused else "four nodes", dt))
    K.BLOCK_FUSED = True
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))


if __name__ == "__main__":
    main()
