"""DeepLab-V3 decoder: ASPP (rates 12/24/36 + image pooling) and the two-output head (main + DSN auxiliary) with
the reference's parameter names (lib/models/modules/decoder_block.py:39-85, 151-179). The only extension is that
the two input widths are arguments (reference hard-codes 1024 / 2048), so ResNet-18/34 encoders fit."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from contrastiveseg_amd.lib.models.tools.module_helper import ClassifierConv1x1, ModuleHelper, SplitConv2d


def _branch(cin, cout, k, rate, bn_type):
    pad = 0 if k == 1 else rate
    return nn.Sequential(SplitConv2d(cin, cout, kernel_size=k, padding=pad, dilation=rate if k == 3 else 1, bias=False),
                         ModuleHelper.BNReLU(cout, bn_type=bn_type))


class ASPPModule(nn.Module):
    def __init__(self, in_dim, out_dim, d_rate=(12, 24, 36), bn_type=None):
        super(ASPPModule, self).__init__()
        self.b0 = _branch(in_dim, out_dim, 1, 1, bn_type)
        self.b1 = _branch(in_dim, out_dim, 3, d_rate[0], bn_type)
        self.b2 = _branch(in_dim, out_dim, 3, d_rate[1], bn_type)
        self.b3 = _branch(in_dim, out_dim, 3, d_rate[2], bn_type)
        self.b4 = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_dim, out_dim, kernel_size=1, padding=0, bias=False),
                                ModuleHelper.BNReLU(out_dim, bn_type=bn_type))
        self.project = nn.Sequential(SplitConv2d(5 * out_dim, out_dim, kernel_size=3, padding=1, bias=False),
                                     ModuleHelper.BNReLU(out_dim, bn_type=bn_type))

    def forward(self, x):
        h, w = x.shape[2:]
        pooled = self.b4(x)
        if pooled.shape[2:] == (1, 1):
            # bilinear upsampling (align_corners=True) of a 1 x 1 map is the constant map, bit for bit; as a broadcast its backward is
            # a sum over the plane instead of upsample_bilinear2d_backward's scatter (12.4 ms per step at 8 x 512 x 65 x 129,
            # profiles/r05_cfg4_step_steady_kernel_stats.csv). Reference: lib/models/modules/decoder_block.py:74-77.
            pooled = pooled.expand(-1, -1, h, w)
        else:
            pooled = F.interpolate(pooled, size=(h, w), mode='bilinear', align_corners=True)
        return self.project(torch.cat((self.b0(x), self.b1(x), self.b2(x), self.b3(x), pooled), dim=1))


class DeepLabHead(nn.Module):
    def __init__(self, num_classes, bn_type=None, in_channels=(1024, 2048)):
        super(DeepLabHead, self).__init__()
        self.layer_dsn = nn.Sequential(SplitConv2d(in_channels[0], 256, kernel_size=3, stride=1, padding=1),
                                       ModuleHelper.BNReLU(256, bn_type=bn_type),
                                       ClassifierConv1x1(256, num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        self.layer_aspp = ASPPModule(in_channels[1], 512, bn_type=bn_type)
        self.refine = nn.Sequential(SplitConv2d(512, 512, kernel_size=3, padding=1, stride=1, bias=False),
                                    ModuleHelper.BatchNorm2d(bn_type=bn_type)(512),
                                    ClassifierConv1x1(512, num_classes, kernel_size=1, stride=1, bias=True))

    def forward(self, x):
        x_dsn = self.layer_dsn(x[2])
        x_seg = self.refine(self.layer_aspp(x[3]))
        return [x_seg, x_dsn]
