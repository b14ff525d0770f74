"""kernels.BasicBlockNative (csrc_host/block_exec.cpp, opt-in: CSEG_NATIVE_BLOCK=1) against kernels.BasicBlockSplit on cuda:0 at a
benched branch shape: same kernels in the same order, so output, gradients and BN buffers must be bit-identical; plus the host time
per block of both routes. Prints one JSON line. Run in its own process by tests/test_zz_gpu_default_routes.py (a crash in native code
must not take the test session down)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import BasicBlock
from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    K.BLOCK_FUSED = True
    if K.native_block_module() is None:
        print(json.dumps({"error": "_cseg_native.so not built"}))
        return
    n_blk = 4
    torch.manual_seed(9)
    net = mark_conv_bn_pairs(torch.nn.Sequential(*[BasicBlock(48, 48, bn_type="torchbn") for _ in range(n_blk)]).to(dev).train())
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    x0 = torch.randn(8, 48, 128, 256, device=dev)
    gy = torch.randn(8, 48, 128, 256, device=dev)
    res, host_us = {}, {}
    for native in (False, True):
        K.NATIVE_BLOCK = native
        net.load_state_dict(state0)
        K.SPLIT_WEIGHTS.invalidate()
        for _ in range(4):
            net.zero_grad()
            x = x0.clone().requires_grad_(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = net(x * 1.0)
            y.backward(gy)
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
        res[native] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()] + [b.clone() for b in net.buffers()]
        host_us[native] = host * 1e6 / n_blk
    same = all(torch.equal(a, b) for a, b in zip(res[False], res[True]))
    print(json.dumps({"bit_identical": bool(same), "host_us_per_block": [round(host_us[False]), round(host_us[True])]}))


if __name__ == "__main__":
    main()
