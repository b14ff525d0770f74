"""Deep-stem ResNet encoders (R-18 / R-50 / R-101) in plain and dilated form, with the reference's parameter
names, construction order and initialisation (lib/models/backbones/resnet/resnet_models.py:107-178,
resnet_backbone.py:21-118) so checkpoints interchange and equal seeds give equal weights. The classifier head
(avgpool + fc) of the reference's ResNet is created only to consume the same random numbers and then dropped,
exactly like the reference's backbone wrappers drop it. Convolutions and max-pool run on MIOpen; BatchNorm with the
ReLU / residual add behind it on the fused cseg_bn_* kernels (lib/models/tools/fused_bn.py)."""
import math
from collections import OrderedDict

import torch.nn as nn

from contrastiveseg_amd.lib.models.tools.module_helper import ModuleHelper, SplitConv2d

LAYERS = {'resnet18': ('basic', [2, 2, 2, 2]), 'resnet34': ('basic', [3, 4, 6, 3]),
          'resnet50': ('bottleneck', [3, 4, 6, 3]), 'resnet101': ('bottleneck', [3, 4, 23, 3]),
          'resnet152': ('bottleneck', [3, 8, 36, 3])}


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_type=None):
        super(BasicBlock, self).__init__()
        bn = ModuleHelper.BatchNorm2d(bn_type=bn_type)
        self.conv1 = SplitConv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = bn(planes)
        self.conv2 = SplitConv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = bn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.bn1(self.conv1(x), relu=True)
        res = x if self.downsample is None else self.downsample(x)
        return self.bn2(self.conv2(out), residual=res, relu=True)     # BN + residual add + ReLU in one apply pass


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_type=None):
        super(Bottleneck, self).__init__()
        bn = ModuleHelper.BatchNorm2d(bn_type=bn_type)
        self.conv1 = SplitConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = bn(planes)
        self.conv2 = SplitConv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = bn(planes)
        self.conv3 = SplitConv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = bn(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        res = x if self.downsample is None else self.downsample(x)
        return self.bn3(self.conv3(out), residual=res, relu=True)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, deep_base=False, bn_type=None):
        super(ResNet, self).__init__()
        bn = ModuleHelper.BatchNorm2d(bn_type=bn_type)
        self.inplanes = 128 if deep_base else 64
        if deep_base:
            # the stateless relu1..3 children of the reference's stem are folded into the norm kernels
            stem = [('conv1', nn.Conv2d(3, 64, 3, 2, 1, bias=False)), ('bn1', bn(64, act='relu')),
                    ('conv2', SplitConv2d(64, 64, 3, 1, 1, bias=False)), ('bn2', bn(64, act='relu')),
                    ('conv3', SplitConv2d(64, 128, 3, 1, 1, bias=False)), ('bn3', bn(128, act='relu'))]
        else:
            stem = [('conv1', nn.Conv2d(3, 64, 7, 2, 3, bias=False)), ('bn1', bn(64, act='relu'))]
        self.resinit = nn.Sequential(OrderedDict(stem))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0], 1, bn_type)
        self.layer2 = self._make_layer(block, 128, layers[1], 2, bn_type)
        self.layer3 = self._make_layer(block, 256, layers[2], 2, bn_type)
        self.layer4 = self._make_layer(block, 512, layers[3], 2, bn_type)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, bn):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride, bn_type):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(SplitConv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 ModuleHelper.BatchNorm2d(bn_type=bn_type)(planes * block.expansion))
        chain = [block(self.inplanes, planes, stride, down, bn_type=bn_type)]
        self.inplanes = planes * block.expansion
        chain += [block(self.inplanes, planes, bn_type=bn_type) for _ in range(1, blocks)]
        return nn.Sequential(*chain)


def _dilate(module, rate):
    """Turn the stride-2 stage into a stride-1 stage with holes (reference resnet_backbone.py:88-101)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            if m.stride == (2, 2):
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation = (rate // 2, rate // 2)
                    m.padding = (rate // 2, rate // 2)
            elif m.kernel_size == (3, 3):
                m.dilation = (rate, rate)
                m.padding = (rate, rate)


class ResnetFeatures(nn.Module):
    """Returns [stem, pool, layer1, layer2, layer3, layer4] like Normal/DilatedResnetBackbone.forward."""

    def __init__(self, net, dilate_scale=None, multi_grid=(1, 2, 4)):
        super(ResnetFeatures, self).__init__()
        self.num_features = 512 * net.layer4[0].expansion
        if dilate_scale == 8:
            _dilate(net.layer3, 2)
            if multi_grid is None:
                _dilate(net.layer4, 4)
            else:
                for i, r in enumerate(multi_grid[:len(net.layer4)]):   # R-18/34 have 2-3 blocks (the reference
                    _dilate(net.layer4[i], int(4 * r))                  # indexes past them and fails)
        elif dilate_scale == 16:
            if multi_grid is None:
                _dilate(net.layer4, 2)
            else:
                for i, r in enumerate(multi_grid[:len(net.layer4)]):
                    _dilate(net.layer4[i], int(2 * r))
        self.resinit, self.maxpool = net.resinit, net.maxpool
        self.layer1, self.layer2, self.layer3, self.layer4 = net.layer1, net.layer2, net.layer3, net.layer4

    def get_num_features(self):
        return self.num_features

    def forward(self, x):
        feats = []
        x = self.resinit(x); feats.append(x)
        x = self.maxpool(x); feats.append(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats


class ResNetBackbone(object):
    """Keys like the reference factory (resnet_backbone.py:121-290): [deepbase_]resnetNN[_dilated8|_dilated16]."""

    def __init__(self, configer):
        self.configer = configer

    def __call__(self):
        arch = self.configer.get('network', 'backbone')
        multi_grid = self.configer.get('network', 'multi_grid') if self.configer.exists('network', 'multi_grid') else None
        name, dil = arch, None
        for tag, d in (('_dilated8', 8), ('_dilated16', 16)):
            if arch.endswith(tag):
                name, dil = arch[:-len(tag)], d
        deep = name.startswith('deepbase_')
        base = name[len('deepbase_'):] if deep else name
        if base not in LAYERS:
            raise Exception('Architecture undefined!')
        kind, layers = LAYERS[base]
        net = ResNet(BasicBlock if kind == 'basic' else Bottleneck, layers, deep_base=deep,
                     bn_type=self.configer.get('network', 'bn_type'))
        net = ModuleHelper.load_model(net, pretrained=self.configer.get('network', 'pretrained'))
        if dil is None:
            return ResnetFeatures(net)
        return ResnetFeatures(net, dilate_scale=dil, multi_grid=multi_grid)
