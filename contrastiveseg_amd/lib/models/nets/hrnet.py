"""HRNet-W48 contrastive segmentors with the reference's registry classes, parameter names and output dicts
(lib/models/nets/hrnet.py:59-95 HRNet_W48_CONTRAST, :98-150 HRNet_W48_OCR_CONTRAST, :153-188 HRNet_W48_MEM).

The 3x F.interpolate + torch.cat that builds the 720-channel head input (:86-91) is one HIP kernel (cseg_upcat_fwd_amax / _bwd, which
also leaves the max|.| record of its result); the head's 3x3 / 1x1 convolutions run on the split-operand kernels (module_helper), the
classifier on csrc/cls1x1.hip with the Dropout2d in front of it folded into its weights (FoldedDropout2d + ClassifierConv1x1)."""
import torch
import torch.nn as nn

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.models.backbones.backbone_selector import BackboneSelector
from contrastiveseg_amd.lib.models.modules.projection import ProjectionHead
from contrastiveseg_amd.lib.models.modules.spatial_ocr_block import SpatialGather_Module, SpatialOCR_Module
from contrastiveseg_amd.lib.models.tools.module_helper import (ClassifierConv1x1, FoldedDropout2d, HeadConv3x3, ModuleHelper,
                                                                  SplitConv2d)


import os as _os

_HEAD_STREAMS = {}
# measured (GPU call r04j17b, after the max|.| record race was fixed): 88.3 / 88.0 ms per step with the heads forked, 86.7 / 88.2
# without -- two chip-filling MFMA kernels next to each other gain nothing; opt-in
HEAD_FORK = _os.environ.get("CSEG_HEAD_FORK", "0") == "1"


def _fork_heads(feats):
    """(current stream, side stream) when the heads may run concurrently: eager GPU steps with the stream forks on
    (hrnet_backbone.EAGER_FORKS); None otherwise (CPU, hipGraph capture, CSEG_BRANCH_STREAMS=0)."""
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    if not (HEAD_FORK and feats.is_cuda and HB.EAGER_FORKS and torch.is_grad_enabled()) or torch.cuda.is_current_stream_capturing():
        return None
    key = feats.device.index
    if key not in _HEAD_STREAMS:
        _HEAD_STREAMS[key] = torch.cuda.Stream(device=feats.device)
    return torch.cuda.current_stream(feats.device), _HEAD_STREAMS[key]


class HRNet_W48_CONTRAST(nn.Module):
    def __init__(self, configer):
        super(HRNet_W48_CONTRAST, self).__init__()
        self.configer = configer
        self.num_classes = self.configer.get('data', 'num_classes')
        self.backbone = BackboneSelector(configer).get_backbone()
        self.proj_dim = self.configer.get('contrast', 'proj_dim')
        in_channels = self.backbone.num_features          # 720 = 48 + 96 + 192 + 384 for W48
        # (constructed in the reference's order -- a seeded initialisation draws the parameters of the 3x3 convolution first)
        conv3x3 = HeadConv3x3(in_channels)
        bnrelu = ModuleHelper.BNReLU(in_channels, bn_type=self.configer.get('network', 'bn_type'))
        classifier = ClassifierConv1x1(in_channels, self.num_classes, kernel_size=1, stride=1, padding=0, bias=False)
        # the dropout mask is folded into the classifier's weights (module_helper.FoldedDropout2d)
        self.cls_head = nn.Sequential(conv3x3, bnrelu, FoldedDropout2d(0.10, classifier), classifier)
        self.proj_head = ProjectionHead(dim_in=in_channels, proj_dim=self.proj_dim)

    def forward(self, x_, with_embed=False, is_eval=False):
        feats = K.upsample_concat(self.backbone(x_))
        fork = _fork_heads(feats)
        if fork is not None:
            # the two heads read the same 720-channel tensor and share nothing else: the projection head (1x1 GEMMs + HBM-bound BN /
            # normalise passes) runs on a side stream under the MFMA-bound 3x3 convolution of the classifier head, and autograd
            # replays the fork in backward
            cur, side = fork
            side.wait_stream(cur)
            feats.record_stream(side)
            with torch.cuda.stream(side):
                emb = self.proj_head(feats)
        out = {'seg': self.cls_head(feats)}
        if out['seg'].is_cuda and self.training and not torch.cuda.is_current_stream_capturing():
            # lets the criterion start anchor mining (and its one host round trip) on a side HIP stream while the
            # projection head below is still running on the compute stream (lib/loss/loss_contrast.py)
            out['seg_ready'] = torch.cuda.Event()
            out['seg_ready'].record()
        if fork is not None:
            cur.wait_stream(side)
            emb.record_stream(cur)
            out['embed'] = emb
        else:
            out['embed'] = self.proj_head(feats)
        return out


class HRNet_W48_OCR_CONTRAST(nn.Module):
    def __init__(self, configer):
        super(HRNet_W48_OCR_CONTRAST, self).__init__()
        self.configer = configer
        self.num_classes = self.configer.get('data', 'num_classes')
        self.backbone = BackboneSelector(configer).get_backbone()
        self.proj_dim = self.configer.get('contrast', 'proj_dim')
        bn_type = self.configer.get('network', 'bn_type')
        in_channels = self.backbone.num_features
        self.conv3x3 = nn.Sequential(SplitConv2d(in_channels, 512, kernel_size=3, stride=1, padding=1),
                                     ModuleHelper.BNReLU(512, bn_type=bn_type))
        self.ocr_gather_head = SpatialGather_Module(self.num_classes)
        self.ocr_distri_head = SpatialOCR_Module(in_channels=512, key_channels=256, out_channels=512, scale=1,
                                                 dropout=0.05, bn_type=bn_type)
        self.cls_head = ClassifierConv1x1(512, self.num_classes, kernel_size=1, stride=1, padding=0, bias=True)
        self.aux_head = nn.Sequential(
            HeadConv3x3(in_channels),
            ModuleHelper.BNReLU(in_channels, bn_type=bn_type),
            ClassifierConv1x1(in_channels, self.num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        self.proj_head = ProjectionHead(dim_in=in_channels, proj_dim=self.proj_dim)

    def forward(self, x_, with_embed=False, is_eval=False):
        feats = K.upsample_concat(self.backbone(x_))
        out_aux = self.aux_head(feats)
        emb = self.proj_head(feats)
        feats = self.conv3x3(feats)
        context = self.ocr_gather_head(feats, out_aux)
        out = self.cls_head(self.ocr_distri_head(feats, context))
        return {'seg': out, 'seg_aux': out_aux, 'embed': emb}


class ContrastMemoryModel(nn.Module):
    """encoder_q + per-class segment / pixel queues as buffers (so they live in the state_dict), the structure of the
    reference's HRNet_W48_MEM (:153-188) with the encoder as a class attribute. The momentum key encoder of the
    reference is dead code (no encoder_k exists, :173-176) and is not carried.

    Subclasses: HRNet_W48_MEM (reference key 'hrnet_w48_mem'); DeepLabV3_MEM and HRNet_W48_OCR_MEM give BASELINE.json
    configs[3] / [4] (DeepLab + pixel memory bank, OCR + region-level memory) a buildable form -- the reference has no
    such registry entries ("configs the reference cannot build as written"); they pair with
    'mem_contrast_auxce_loss'. The segment queue (per-image class-mean embeddings) is the region-level memory."""
    ENCODER = None

    def __init__(self, configer, dim=None, m=0.999, with_masked_ppm=False):
        super(ContrastMemoryModel, self).__init__()
        self.configer = configer
        self.m = m
        self.r = self.configer.get('contrast', 'memory_size')
        self.with_masked_ppm = with_masked_ppm
        num_classes = self.configer.get('data', 'num_classes')
        if dim is None:
            dim = self.configer.get('contrast', 'proj_dim') if self.configer.exists('contrast', 'proj_dim') else 256
        self.encoder_q = type(self).ENCODER(configer)
        self.register_buffer("segment_queue", torch.randn(num_classes, self.r, dim))
        self.segment_queue = nn.functional.normalize(self.segment_queue, p=2, dim=2)
        self.register_buffer("segment_queue_ptr", torch.zeros(num_classes, dtype=torch.long))
        self.register_buffer("pixel_queue", torch.randn(num_classes, self.r, dim))
        self.pixel_queue = nn.functional.normalize(self.pixel_queue, p=2, dim=2)
        self.register_buffer("pixel_queue_ptr", torch.zeros(num_classes, dtype=torch.long))

    def forward(self, im_q, lb_q=None, with_embed=True, is_eval=False):
        if is_eval is True or lb_q is None:
            return self.encoder_q(im_q, with_embed=with_embed)
        ret = self.encoder_q(im_q)
        q = ret['embed']
        ret.update({'key': q.detach(), 'lb_key': lb_q.detach()})
        return ret


class HRNet_W48_MEM(ContrastMemoryModel):
    """reference lib/models/nets/hrnet.py:153-188 (dim defaults to 256 there)"""
    ENCODER = HRNet_W48_CONTRAST

    def __init__(self, configer, dim=256, m=0.999, with_masked_ppm=False):
        super(HRNet_W48_MEM, self).__init__(configer, dim=dim, m=m, with_masked_ppm=with_masked_ppm)


class HRNet_W48_OCR_MEM(ContrastMemoryModel):
    ENCODER = HRNet_W48_OCR_CONTRAST
