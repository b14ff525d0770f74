"""Memory-bank variant: PixelContrastLoss / ContrastCELoss of lib/loss/loss_contrast_mem.py:15-231.

Differences from the bank-free loss (lib/loss/loss_contrast.py of this package) follow the reference:
the contrast set is `cat(segment_queue, pixel_queue, dim=1)` with class 0 skipped and the last 2*ms rows left
zero with label 0 (:91-105), `contrast_count = 1`, and the self mask still removes column i of row i although
the columns are bank entries (:134-138). The bank is read IN PLACE by cseg_contrast_fwd/bwd (mode 2): no
[K*2*ms, D] copy and no N x M temporaries besides the similarity workspace. Anchors get gradients, the bank does
not. Every rank contrasts its own anchors against its own copy of the bank, as in the reference."""
from abc import ABC

import numpy as np
import torch
import torch.nn as nn

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.loss.loss_contrast import PixelContrastLoss as _SelfPixelContrastLoss
from contrastiveseg_amd.lib.loss.loss_contrast import _counts_to_host, _grad_slot
from contrastiveseg_amd.lib.loss.loss_helper import FSAuxCELoss, FSCELoss
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class PixelContrastLoss(_SelfPixelContrastLoss):
    def forward(self, feats, labels=None, predict=None, queue=None, seg=None, segment_queue=None,
                pixel_queue=None, seg_ready=None):
        """Reference signature is (feats, labels, predict, queue) with queue = cat(segment, pixel) [K, 2*ms, D]
        (:154, :221). Passing the two queues separately avoids that concat."""
        if queue is not None and segment_queue is None:
            ms = queue.shape[1] // 2
            segment_queue = queue[:, :ms].contiguous()
            pixel_queue = queue[:, ms:].contiguous()
        if segment_queue is None:
            return super(PixelContrastLoss, self).forward(feats, labels, predict=predict, seg=seg, seg_ready=seg_ready)
        B, Dm, h, w = feats.shape
        P = h * w
        cp = self._mine(feats, labels, predict, seg, seg_ready)
        plan = self._plan(_counts_to_host(cp))
        dev = feats.device
        sel_pos = torch.from_numpy(plan.row_img.astype(np.int32) * P + plan.row_off).to(dev, non_blocking=True)
        a_lab = torch.from_numpy(plan.row_lab.astype(np.int32)).to(dev, non_blocking=True)
        loss, sel_pix = K.PixelContrast.apply(feats, cp["part_idx"], sel_pos, a_lab, "bank", self.temperature,
                                              self.base_temperature, segment_queue.contiguous(),
                                              pixel_queue.contiguous(), _grad_slot(feats))
        self.last_selection = {"sel_pix": sel_pix, "plan": plan}
        return loss


class ContrastCELoss(nn.Module, ABC):
    def __init__(self, configer=None):
        super(ContrastCELoss, self).__init__()
        self.configer = configer
        ignore_index = -1
        if self.configer.exists('loss', 'params') and 'ce_ignore_index' in self.configer.get('loss', 'params'):
            ignore_index = self.configer.get('loss', 'params')['ce_ignore_index']
        Log.info('ignore_index: {}'.format(ignore_index))
        self.loss_weight = self.configer.get('contrast', 'loss_weight')
        self.use_rmi = self.configer.get('contrast', 'use_rmi')
        self.use_lovasz = self.configer.get('contrast', 'use_lovasz') \
            if self.configer.exists('contrast', 'use_lovasz') else False
        if self.use_rmi or self.use_lovasz:
            raise NotImplementedError("contrast.use_rmi / use_lovasz criteria are outside the accelerated hot path")
        self.seg_criterion = FSCELoss(configer=configer)
        self.contrast_criterion = PixelContrastLoss(configer=configer)

    def forward(self, preds, target, with_embed=False):
        assert "seg" in preds
        assert "embed" in preds
        seg = preds['seg']
        embedding = preds['embed']
        segment_queue = preds.get('segment_queue')
        pixel_queue = preds.get('pixel_queue')
        loss = self.seg_criterion(seg, target)
        if segment_queue is not None and pixel_queue is not None:
            loss_contrast = self.contrast_criterion(embedding, target, seg=seg, segment_queue=segment_queue,
                                                    pixel_queue=pixel_queue, seg_ready=preds.get('seg_ready'))
        else:
            loss_contrast = 0
        # the two terms of the last call, detached (no host sync): the segmentation term is a smooth function of the weights, the
        # contrastive term is not (argmax decides hard / easy, rounding-level changes of the logits move anchors between the sets) --
        # tests that compare two implementations after an SGD step bound the former tightly and the latter loosely
        self.last_terms = (loss.detach(), loss_contrast.detach())
        if with_embed is True:
            return loss + self.loss_weight * loss_contrast
        return loss + 0 * loss_contrast


class ContrastAuxCELoss(ContrastCELoss):
    """FSAuxCELoss([seg_aux, seg]) + the memory-bank contrast term: what loss_contrast_mem.py:234-276 of the reference
    sets out to be. As written there it is unregistered, reads the key 'embedding' (the models emit 'embed'), never
    receives the queues and names an un-imported criterion; here it takes the same `preds` dict
    as the registered memory criterion plus 'seg_aux', registered as 'mem_contrast_auxce_loss' for the DeepLab / OCR
    memory models (BASELINE.json configs[3] / [4])."""

    def __init__(self, configer=None):
        super(ContrastAuxCELoss, self).__init__(configer)
        self.seg_criterion = FSAuxCELoss(configer=configer)

    def forward(self, preds, target, with_embed=False):
        assert "seg" in preds
        assert "seg_aux" in preds
        assert "embed" in preds
        seg = preds['seg']
        loss = self.seg_criterion([preds['seg_aux'], seg], target)
        segment_queue = preds.get('segment_queue')
        pixel_queue = preds.get('pixel_queue')
        if segment_queue is not None and pixel_queue is not None:
            loss_contrast = self.contrast_criterion(preds['embed'], target, seg=seg, segment_queue=segment_queue,
                                                    pixel_queue=pixel_queue, seg_ready=preds.get('seg_ready'))
        else:
            loss_contrast = 0
        # the two terms of the last call, detached (no host sync): the segmentation term is a smooth function of the weights, the
        # contrastive term is not (argmax decides hard / easy, rounding-level changes of the logits move anchors between the sets) --
        # tests that compare two implementations after an SGD step bound the former tightly and the latter loosely
        self.last_terms = (loss.detach(), loss_contrast.detach())
        if with_embed is True:
            return loss + self.loss_weight * loss_contrast
        return loss + 0 * loss_contrast
