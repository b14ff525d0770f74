"""Norm plug-in of the reference (lib/models/tools/module_helper.py:29-121), restricted to the two branches the
hot-path configs reach: 'torchbn' -> FusedBatchNorm2d (an nn.BatchNorm2d), 'torchsyncbn' -> FusedSyncBatchNorm (an
nn.SyncBatchNorm: global-batch statistics over RCCL when a process group exists, plain batch norm otherwise). Both are
computed by the cseg_bn_* HIP kernels and can fuse the ReLU / residual add that follows (fused_bn.py); parameters,
buffers and state_dict keys are those of the torch classes. Pretrained loading mirrors :124-235 for the two backbone
families of the hot path."""
import torch
import torch.nn as nn

from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d, FusedSyncBatchNorm
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

_NORMS = {'torchbn': FusedBatchNorm2d, 'torchsyncbn': FusedSyncBatchNorm}


class Conv3x3(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) for the bias-free 3x3, stride-1, pad-1 convolutions of the residual
    branches. Routes, in this order, for shapes whose launch fills the chip:
      * kernels.CONV3X3_SPLIT_BF16 (default) and a channel count in kernels.CONV3X3_SB_BRANCH_CHANNELS (48 / 96 / 192):
        forward and backward-data on the split-bf16 MFMA kernel (csrc/conv3x3_sb.hip), weight gradient on the split-bf16
        kernel (48 / 96 at widths % 64), the fp32-MFMA kernel or MIOpen (kernels.Conv3x3SplitBF16);
      * 48 / 96 channels otherwise: the fp32-MFMA kernel (csrc/conv3x3.hip: 121 vs 167 us and 113 vs 119 us per forward
        against MIOpen at the benched shapes, tools/conv3x3_probe.py);
      * everything else: the reference's nn.Conv2d on MIOpen."""
    MFMA_CHANNELS = (48, 96)
    # set by the owner when a BatchNorm consumes the output directly (conv -> bn chains, hrnet_backbone.py:49-65 of the reference): the
    # split kernels then produce the BN statistics in their epilogue (csrc/cseg_stats.h) and fused_bn skips its statistics pass
    bn_follows = False

    def __init__(self, inplanes, planes, stride=1):
        super(Conv3x3, self).__init__(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        pair = (self.in_channels, self.out_channels) in K.CONV3X3_SB_WRW_PAIRS        # transition 1: 256 -> 48
        if K._on_device(x) and self.stride == (1, 1) and self.dilation == (1, 1) and (self.in_channels == self.out_channels or pair):
            if (K.CONV3X3_SPLIT_BF16 and (pair or self.in_channels in K.CONV3X3_SB_BRANCH_CHANNELS)
                    and K.conv3x3_sb_eligible(x, self.weight)
                    and K.conv3x3_sb_tiles(x, self.out_channels) >= K.CONV3X3_SB_MIN_TILES):
                return K.conv3x3_split_bf16(x, self.weight, None, self.bn_follows)
            if not pair and self.in_channels in self.MFMA_CHANNELS and K.conv3x3_eligible(x, self.weight):
                return K.conv3x3(x, self.weight)
        if (K._on_device(x) and self.stride == (2, 2) and self.dilation == (1, 1)
                and (K.conv3x3_s2_fwd_eligible(x, self.weight) or K.conv3x3_s2_wrw_eligible(x, self.weight))):
            # downsampling convolutions of the fuse / transition layers: csrc/conv3x3_s2.hip + the stride-2 weight gradient
            return K.conv3x3_s2_split(x, self.weight, self.bn_follows)
        return super(Conv3x3, self).forward(x)

    def forward_fork(self, x):
        """(self(x), x) for the first convolution of a residual block whose identity path is `x` itself: where the split kernels take
        this layer, both come out of ONE autograd node whose backward adds the identity path's gradient in the epilogue of the
        backward-data kernel (kernels.Conv3x3SplitFork); otherwise plainly (self(x), x)."""
        from contrastiveseg_amd import kernels as K
        if (K.CONV3X3_FORK and K._on_device(x) and x.requires_grad and self.stride == (1, 1) and self.dilation == (1, 1)
                and self.in_channels == self.out_channels and K.CONV3X3_SPLIT_BF16
                and self.in_channels in K.CONV3X3_SB_BRANCH_CHANNELS and K.conv3x3_sb_eligible(x, self.weight)
                and K.conv3x3_sb_tiles(x, self.out_channels) >= K.CONV3X3_SB_MIN_TILES):
            return K.conv3x3_split_fork(x, self.weight, self.bn_follows)
        return self.forward(x), x


class StemConv3x3(nn.Conv2d):
    """nn.Conv2d(3, planes, 3, 2, 1, bias=False) -- the first convolution of the stem (reference hrnet_backbone.py:516): same parameters
    and state_dict; on the GPU, for planes = 64 and an input that needs no gradient, the fp32 stem kernels of csrc/conv3x3_stem.hip
    (forward and weight gradient), else the reference's convolution on MIOpen."""

    def __init__(self, planes):
        super(StemConv3x3, self).__init__(3, planes, kernel_size=3, stride=2, padding=1, bias=False)

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        if K.conv3x3_s2_rgb_eligible(x, self.weight):
            return K.conv3x3_s2_rgb(x, self.weight)
        return super(StemConv3x3, self).forward(x)


class HeadConv3x3(nn.Conv2d):
    """nn.Conv2d(C, C, 3, 1, 1) with bias (same parameters / state_dict) for the 720 -> 720 convolution in front of the
    classifier (44 % of the forward FLOPs of HRNet-W48-contrast). With kernels.CONV3X3_SPLIT_BF16 on, forward and
    backward-data run on the split-bf16 MFMA kernel (csrc/conv3x3_sb.hip: 11.0 vs 19.8 ms per direction at bs 8,
    128x256); otherwise, and for shapes it does not cover, this is the reference's nn.Conv2d on MIOpen."""

    bn_follows = False        # see Conv3x3

    def __init__(self, channels):
        super(HeadConv3x3, self).__init__(channels, channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        if (K._on_device(x) and K.CONV3X3_SPLIT_BF16 and K.conv3x3_sb_eligible(x, self.weight)
                and K.conv3x3_sb_tiles(x, self.out_channels) >= K.CONV3X3_SB_MIN_TILES):
            return K.conv3x3_split_bf16(x, self.weight, self.bias, self.bn_follows)
        return super(HeadConv3x3, self).forward(x)


class ClassifierConv1x1(nn.Conv2d):
    """nn.Conv2d(cin, num_classes, 1) (same parameters / state_dict): the classifier at the end of `cls_head` / the auxiliary head
    (reference lib/models/nets/hrnet.py:73-80, :113-131). Up to 32 classes on the GPU: the three fp32 streaming kernels of
    csrc/cls1x1.hip (kernels.cls1x1) instead of library GEMMs with 19 columns; a channel mask left on the input by FoldedDropout2d is
    folded into the weights. Anything else (more classes, CPU) is the reference's convolution -- with the mask applied first."""

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        fold = getattr(x, '_cseg_drop_mask', None)
        mask = fold[0] if fold is not None and fold[1] == x._version else None
        if (K.cls1x1_eligible(x, self.weight) and self.stride == (1, 1) and self.padding == (0, 0) and self.dilation == (1, 1)
                and self.groups == 1):
            return K.cls1x1(x, self.weight, self.bias, mask)
        if mask is not None:
            x = x * mask
        return super(ClassifierConv1x1, self).forward(x)


class FoldedDropout2d(nn.Dropout2d):
    """nn.Dropout2d in front of a ClassifierConv1x1 (`consumer`): in training it draws the channel mask exactly as F.dropout2d does
    (aten/src/ATen/native/Dropout.cpp, feature noise: `input.new_empty([B, C, 1, 1]).bernoulli_(1 - p).div_(1 - p)` -- the same
    generator draws, so a run stays comparable with the reference's) but does NOT multiply the activation: the mask travels with the
    tensor and the classifier multiplies its 19 x 720 weights instead (one read + one write of the 755 MB activation less, forward
    and backward). Where the classifier's fast path does not apply this is nn.Dropout2d."""

    def __init__(self, p, consumer):
        super(FoldedDropout2d, self).__init__(p)
        self._consumer = (consumer,)               # (a tuple: not registered as a sub-module -- the Sequential already owns it)

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        conv = self._consumer[0]
        if (not self.training or self.inplace or not 0.0 < self.p < 1.0 or x.dim() != 4 or not isinstance(conv, ClassifierConv1x1)
                or not K.cls1x1_eligible(x, conv.weight)):
            return super(FoldedDropout2d, self).forward(x)
        mask = x.new_empty((x.shape[0], x.shape[1], 1, 1)).bernoulli_(1 - self.p).div_(1 - self.p)
        out = x.view_as(x)                         # a new tensor object for the attribute; the values are untouched
        out._cseg_drop_mask = (mask, out._version)
        return out


class Conv1x1(nn.Conv2d):
    """nn.Conv2d(cin, cout, 1) (same parameters / state_dict). With kernels.CONV1X1_SPLIT_BF16 on, forward and
    backward-data of the shapes the split-bf16 kernel covers (csrc/conv1x1_sb.hip) run there -- and the weight gradient
    with kernels.CONV1X1_SB_WRW -- otherwise this is the reference's pointwise convolution on rocBLAS / MIOpen."""

    bn_follows = False        # see Conv3x3

    def __init__(self, cin, cout, bias=True):
        super(Conv1x1, self).__init__(cin, cout, kernel_size=1, bias=bias)

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        if (K._on_device(x) and K.CONV1X1_SPLIT_BF16 and K.conv1x1_sb_eligible(x, self.weight)
                and K.conv1x1_sb_tiles(x, self.out_channels) >= K.CONV1X1_SB_MIN_TILES):
            return K.conv1x1_split_bf16(x, self.weight, self.bias, self.bn_follows)
        return super(Conv1x1, self).forward(x)

    def forward_skip(self, x):
        """-> (self(x), x for the block's skip connection). On the split kernels in a differentiable pass the pair is ONE autograd node
        whose backward adds the skip gradient in the epilogue of the backward-data kernel (kernels.Conv1x1SplitSkip); otherwise plainly
        (self(x), x)."""
        from contrastiveseg_amd import kernels as K
        import torch
        if (K.SKIP_ADD_FUSED and torch.is_grad_enabled() and x.requires_grad and K._on_device(x) and K.CONV1X1_SPLIT_BF16
                and K.conv1x1_sb_eligible(x, self.weight) and K.conv1x1_sb_tiles(x, self.out_channels) >= K.CONV1X1_SB_MIN_TILES):
            return K.conv1x1_split_skip(x, self.weight, self.bias, self.bn_follows)
        return self(x), x


class SplitConv2d(nn.Conv2d):
    """nn.Conv2d (same constructor, parameters, initialisation and state_dict) for the 1x1 and plain 3x3 convolutions OUTSIDE the HRNet
    branches (round 5): the bottleneck 1x1 layers and the deep stem of the ResNet encoders, ASPP's 1x1 branch and 3x3 projection,
    the DSN / refine convolutions of the DeepLab head, the 3x3 + 1x1 layers of the OCR head (reference
    lib/models/backbones/resnet/resnet_models.py:60-105, lib/models/modules/decoder_block.py:39-85, 151-179,
    lib/models/modules/spatial_ocr_block.py:116-217, lib/models/nets/hrnet.py:113-131). Stride 1, no dilation, groups 1, `same`
    padding and channel counts the split-operand kernels tile (multiples of 48 or 64 both ways), launches that fill the chip: forward
    and backward-data on cseg_conv1x1_split_* / cseg_conv3x3_split_* at ANY width (the 65 x 129 and 130 x 130 maps of these models are
    why the kernels lost their width % 4 requirement), weight gradients there too where kernels.conv*_wrw_wanted says so. Everything
    else (dilated, strided, 7x7, class-count outputs) is the reference's convolution on MIOpen."""

    bn_follows = False        # see Conv3x3

    def forward(self, x):
        from contrastiveseg_amd import kernels as K
        k = self.kernel_size[0]
        if (K._on_device(x) and self.kernel_size in ((1, 1), (3, 3)) and self.stride == (1, 1) and self.dilation == (1, 1)
                and self.groups == 1 and self.padding == (k // 2, k // 2) and self.padding_mode == 'zeros' and x.dim() == 4):
            if k == 1:
                if (K.CONV1X1_SPLIT_BF16 and K.conv1x1_sb_eligible(x, self.weight)
                        and K.conv1x1_sb_tiles(x, self.out_channels) >= K.CONV1X1_SB_MIN_TILES):
                    return K.conv1x1_split_bf16(x, self.weight, self.bias, self.bn_follows)
            elif (K.CONV3X3_SPLIT_BF16 and K.conv3x3_sb_eligible(x, self.weight)
                    and K.conv3x3_sb_tiles(x, self.out_channels) >= K.CONV3X3_SB_MIN_TILES):
                return K.conv3x3_split_bf16(x, self.weight, self.bias, self.bn_follows)
        elif (K._on_device(x) and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.dilation in ((2, 2), (4, 4))
                and self.padding == self.dilation and self.groups == 1 and self.padding_mode == 'zeros' and x.dim() == 4
                and K.conv3x3_dil_eligible(x, self.weight, self.dilation)
                and K.conv3x3_sb_tiles(x, self.out_channels) >= K.CONV3X3_SB_MIN_TILES):
            # layer3 / layer4 of the dilated ResNets (rate 2 / 4): conv3x3_sb16d_kernel
            return K.conv3x3_dil_split(x, self.weight, self.bias, self.dilation[0], self.bn_follows)
        return super(SplitConv2d, self).forward(x)


def mark_conv_bn_pairs(module):
    """Sets `bn_follows` on every split-kernel convolution of `module` whose output goes straight into a BatchNorm: the (conv, norm)
    neighbours of every nn.Sequential (the `_conv_bn` chains, classifier / projection heads, `BNReLU` wrappers included) and the
    conv_k / bn_k attribute pairs of the residual blocks. Called once by the model constructors; purely a performance hint -- a
    convolution marked by mistake only writes a small statistics buffer nobody reads."""
    from contrastiveseg_amd.lib.models.tools.fused_bn import _FusedMixin

    def first_norm(m):
        if isinstance(m, _FusedMixin):
            return True
        return isinstance(m, nn.Sequential) and len(m) > 0 and first_norm(m[0])

    kinds = (Conv3x3, HeadConv3x3, Conv1x1, SplitConv2d)
    for m in module.modules():
        if isinstance(m, nn.Sequential):
            kids = list(m)
            for a, b in zip(kids, kids[1:]):
                if isinstance(a, kinds) and first_norm(b):
                    a.bn_follows = True
        for k in (1, 2, 3):
            conv, bn = getattr(m, 'conv%d' % k, None), getattr(m, 'bn%d' % k, None)
            if isinstance(conv, kinds) and isinstance(bn, _FusedMixin):
                conv.bn_follows = True
    return module


class ModuleHelper(object):
    @staticmethod
    def BatchNorm2d(bn_type='torch', ret_cls=False):
        if bn_type not in _NORMS:
            Log.error('Not support BN type: {}.'.format(bn_type))
            exit(1)
        return _NORMS[bn_type]

    @staticmethod
    def BNReLU(num_features, bn_type=None, **kwargs):
        # reference: nn.Sequential(BN, nn.ReLU()); the ReLU has no state, so folding it into the norm kernel keeps the
        # state_dict keys ('0.weight', ...) unchanged
        return nn.Sequential(ModuleHelper.BatchNorm2d(bn_type)(num_features, act='relu', **kwargs))

    @staticmethod
    def load_model(model, pretrained=None, all_match=True, network='resnet101'):
        if pretrained is None:
            return model
        Log.info('Loading pretrained model:{}'.format(pretrained))
        src = torch.load(pretrained, map_location='cpu')
        dst = model.state_dict()
        if all_match:
            # torchvision-style stems are stored without the `resinit.` prefix (reference :130-139)
            load = {('resinit.' + k if 'resinit.' + k in dst else k): v for k, v in src.items()}
            model.load_state_dict(load)
        else:
            if network != 'hrnet':
                raise NotImplementedError('pretrained loading for {!r} is outside the hot path'.format(network))
            load = {k: v for k, v in src.items() if k in dst}
            Log.info('Missing keys: {}'.format(list(set(dst) - set(load))))
            dst.update(load)
            model.load_state_dict(dst)
        return model
