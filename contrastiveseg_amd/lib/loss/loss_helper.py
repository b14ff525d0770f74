"""Segmentation criteria on the hot path: FSCELoss / FSAuxCELoss with the reference's constructor contract
(lib/loss/loss_helper.py:169-212, 301-313), computed by the fused upsample+CE HIP kernel
(cseg_upsample_ce_fwd/bwd) instead of F.interpolate + nn.CrossEntropyLoss.

`inputs` may be at label resolution (what the reference passes) or coarser: the kernel interpolates on the fly
with bilinear(align_corners=True), which is the identity when the sizes match."""
import torch
import torch.nn as nn

from contrastiveseg_amd import kernels as K


def _ce_params(configer):
    weight, reduction, ignore_index = None, "elementwise_mean", -1
    if configer.exists("loss", "params"):
        p = configer.get("loss", "params")
        if "ce_weight" in p:
            weight = torch.tensor(p["ce_weight"], dtype=torch.float32)
        if "ce_reduction" in p:
            reduction = p["ce_reduction"]
        if "ce_ignore_index" in p:
            ignore_index = p["ce_ignore_index"]
    return weight, reduction, ignore_index


class FSCELoss(nn.Module):
    def __init__(self, configer=None):
        super(FSCELoss, self).__init__()
        self.configer = configer
        weight, reduction, ignore_index = _ce_params(configer)
        if reduction not in ("elementwise_mean", "mean"):
            raise NotImplementedError("ce_reduction %r: only the mean reduction of the shipped configs is "
                                      "implemented on the HIP path" % reduction)
        self.register_buffer("weight", weight, persistent=False)
        self.register_buffer("status", torch.zeros(4, dtype=torch.int32), persistent=False)
        self.ignore_index = ignore_index

    def _one(self, inp, target):
        return K.upsample_ce(inp, target, self.weight, self.ignore_index, self.status)

    def bad_label_count(self, reset=True):
        """Labels seen since the last call that are neither `ce_ignore_index` nor a class id: the kernel drops them
        (nn.CrossEntropyLoss of the reference would assert). One D2H copy: call it where the host syncs anyway
        (Trainer._display does, and raises)."""
        n = int(self.status[1])
        if reset:
            self.status.zero_()
        return n

    def forward(self, inputs, *targets, weights=None, **kwargs):
        if isinstance(inputs, (tuple, list)):
            if weights is None:
                weights = [1.0] * len(inputs)
            loss = 0.0
            for i, inp in enumerate(inputs):
                tgt = targets[i] if len(targets) > 1 else targets[0]
                loss = loss + weights[i] * self._one(inp, tgt)
            return loss
        return self._one(inputs, targets[0])


class FSAuxCELoss(nn.Module):
    """seg_loss * w_seg + aux_loss * w_aux (reference :301-313)."""

    def __init__(self, configer=None):
        super(FSAuxCELoss, self).__init__()
        self.configer = configer
        self.ce_loss = FSCELoss(self.configer)

    def forward(self, inputs, targets, **kwargs):
        aux_out, seg_out = inputs
        seg_loss = self.ce_loss(seg_out, targets)
        aux_loss = self.ce_loss(aux_out, targets)
        lw = self.configer.get("network", "loss_weights")
        return lw["seg_loss"] * seg_loss + lw["aux_loss"] * aux_loss
