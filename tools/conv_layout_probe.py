"""Round-2 probe: every distinct convolution of the benched HRNet-W48 step (bs8, 3x512x1024), timed per direction
(forward, backward-data, backward-weight) through PyTorch/MIOpen with NCHW and with channels_last tensors, weighted by
how often the network uses it. Decides whether the hand-written kernels should go NHWC."""
import collections
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def conv_configs(B=8, H=512, W=1024):
    from contrastiveseg_amd.lib.models.model_manager import ModelManager
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    cfg = Configer(config_dict={"data": {"num_classes": 19},
                                "network": {"backbone": "hrnet48", "model_name": "hrnet_w48_contrast",
                                            "bn_type": "torchbn", "resume": None, "pretrained": None},
                                "contrast": {"proj_dim": 256}})
    net = ModelManager(cfg).semantic_segmentor().to("meta")
    seen = collections.Counter()

    def hook(m, inp, out):
        x = inp[0]
        seen[(tuple(x.shape), m.out_channels, m.kernel_size[0], m.stride[0], m.dilation[0], m.bias is not None)] += 1
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.register_forward_hook(hook)
    import contrastiveseg_amd.lib.models.nets.hrnet as nh
    import contrastiveseg_amd.lib.models.backbones.hrnet_backbone as hb
    from oracle import cpu_port
    nh.K = hb.K = cpu_port          # shape tracing on the meta device only
    with torch.no_grad():
        net(torch.empty(B, 3, H, W, device="meta"))
    return seen


def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_conv(shape, cout, k, stride, dil, cl, iters=4):
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, device="cuda")
    w = torch.randn(cout, C, k, k, device="cuda") * 0.01
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    w.requires_grad_(True)
    pad = dil * (k - 1) // 2
    y = F.conv2d(x, w, None, stride, pad, dil)
    g = torch.randn_like(y)
    torch.autograd.grad(y, (x, w), g, retain_graph=True)          # first use: MIOpen find
    f = ev(lambda: F.conv2d(x, w, None, stride, pad, dil), iters)
    bd = ev(lambda: torch.autograd.grad(y, x, g, retain_graph=True), iters) if C > 3 else 0.0
    bw = ev(lambda: torch.autograd.grad(y, w, g, retain_graph=True), iters)
    return f, bd, bw


def main():
    import contrastiveseg_amd  # noqa: F401  (activates the shipped MIOpen records)
    torch.backends.cudnn.benchmark = False
    cfgs = conv_configs()
    tot = {0: [0.0, 0.0, 0.0], 1: [0.0, 0.0, 0.0]}
    rows = []
    for (shape, cout, k, stride, dil, bias), cnt in sorted(cfgs.items(), key=lambda kv: -kv[1]):
        r = {"shape": shape, "cout": cout, "k": k, "stride": stride, "count": cnt}
        for cl in (0, 1):
            f, bd, bw = time_conv(shape, cout, k, stride, dil, cl)
            r["cl%d" % cl] = [round(f, 3), round(bd, 3), round(bw, 3)]
            for i, v in enumerate((f, bd, bw)):
                tot[cl][i] += v * cnt
        rows.append(r)
        print(json.dumps(r), flush=True)
    print(json.dumps({"total_ms_per_step": {"nchw(fwd,bwd_data,bwd_weight)": [round(v, 1) for v in tot[0]],
                                            "channels_last": [round(v, 1) for v in tot[1]]}}))


if __name__ == "__main__":
    main()
