"""PixelContrastLoss / ContrastCELoss / ContrastAuxCELoss with the reference's constructor and forward contracts
(lib/loss/loss_contrast.py:15-234), computed by the HIP kernels of libcseg_hip.so:

  reference step (loss_contrast.py)                       here
  ------------------------------------------------------  --------------------------------------------------
  :180-181 F.interpolate(seg) + FSCELoss                  cseg_upsample_ce_fwd/bwd (no [B,K,H,W] tensor)
  :183 torch.max(seg,1); :131-134 nearest label resize;   cseg_classify_partition (one pass over seg, stable
  :35-64 unique / nonzero per (image,class)               hard/easy partition in .nonzero() order)
  :66-82 keep rule + torch.randperm                       anchor_sampling.plan_selection (host, same CPU RNG
                                                          stream => bit-identical indices)
  :141-142 NHWC copy, :85-87 gather                       cseg_gather_anchors straight from NCHW
  :91-128 _contrastive                                    cseg_contrast_fwd/bwd (fp32 MFMA)

Data-parallel (one process per GPU, RCCL). Two modes, chosen by `contrast.cross_rank`:
  * false (the code default = the reference's DDP behaviour): every rank contrasts only the <= max_samples anchors
    mined from its own images; nothing but DDP's gradient all-reduce crosses ranks.
  * true (opt-in; set by the shipped HRNet configs because BASELINE.json's multi-GPU configuration asks for it): the
    contrast set is the union of every rank's anchors: counts are all-gathered so all ranks derive the same global
    selection from the same RNG stream, each rank gathers its own rows, rows are all-gathered, and every rank
    evaluates the global loss while back-propagating only into its own embeddings (gradient scaled by world_size so
    that DDP's gradient averaging reproduces the single-process gradient of the global loss). This CHANGES the
    objective relative to the reference's DDP run (world x more negatives per anchor); its oracle is the reference's
    single-process loss on the concatenated global batch. The anchor budget of the global set is
    max_samples * world_size by default (`contrast.cross_rank_budget`: 'per_rank' | 'global').
The memory-bank criterion (loss_contrast_mem.py) always contrasts a rank's own anchors against its own copy of the
bank, as the reference does; `cross_rank` does not apply to it."""
from abc import ABC

import numpy as np
import torch
import torch.nn as nn

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.loss.anchor_sampling import plan_selection
from contrastiveseg_amd.lib.loss.loss_helper import FSAuxCELoss, FSCELoss
from contrastiveseg_amd.lib.utils import distributed as D
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class _GradScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def _counts_to_host(cp):
    """One D2H copy (the only host sync of the loss): counts + status."""
    if "host_counts" in cp:
        flat, done = cp["host_counts"]
        done.synchronize()                       # waits for the side stream only
    else:
        flat = torch.cat([cp["counts"].reshape(-1), cp["status"]]).cpu()
    counts = flat[:-4].reshape(cp["counts"].shape).numpy()
    if int(flat[-4]) != 0:
        raise RuntimeError("PixelContrastLoss: %d label values are neither ignore_label nor in [0, num_classes); "
                           "the HIP mining kernel handles classes 0..K-1 only" % int(flat[-4]))
    return counts


def _grad_slot(embed):
    """kernels.SparseGradSlot the projection head attached to the embedding it produced (row-sparse backward, opt-in:
    lib/models/modules/projection.py), or None = the dense zero-filled gradient."""
    return getattr(embed, '_cseg_grad_slot', None)


class PixelContrastLoss(nn.Module, ABC):
    def __init__(self, configer):
        super(PixelContrastLoss, self).__init__()
        self.configer = configer
        self.temperature = self.configer.get('contrast', 'temperature')
        self.base_temperature = self.configer.get('contrast', 'base_temperature')
        self.ignore_label = -1
        if self.configer.exists('loss', 'params') and 'ce_ignore_index' in self.configer.get('loss', 'params'):
            self.ignore_label = self.configer.get('loss', 'params')['ce_ignore_index']
        self.max_samples = self.configer.get('contrast', 'max_samples')
        self.max_views = self.configer.get('contrast', 'max_views')
        self.cross_rank = False      # reference parity unless the config opts in (see the module docstring)
        if self.configer.exists('contrast', 'cross_rank'):
            self.cross_rank = bool(self.configer.get('contrast', 'cross_rank'))
        # anchor budget of the cross-rank set: 'per_rank' = max_samples * world_size anchors in total (every rank of
        # the reference's DDP run owns a max_samples budget), 'global' = max_samples in total (parity with a
        # single process that sees the concatenated global batch).
        self.cross_rank_budget = 'per_rank'
        if self.configer.exists('contrast', 'cross_rank_budget'):
            self.cross_rank_budget = self.configer.get('contrast', 'cross_rank_budget')
        assert self.cross_rank_budget in ('per_rank', 'global')
        # RNG of the cross-rank set: 'local' = every rank draws only its own segments from its own generator (host
        # cost independent of world size); 'global' = every rank draws for all global segments in global image order,
        # which reproduces a single process on the concatenated batch index for index (host cost grows with world).
        self.cross_rank_rng = 'local'
        if self.configer.exists('contrast', 'cross_rank_rng'):
            self.cross_rank_rng = self.configer.get('contrast', 'cross_rank_rng')
        assert self.cross_rank_rng in ('local', 'global')
        self._side = None            # side HIP stream for mining (created lazily on the first GPU call)
        self.last_selection = None   # {'sel_pix': i32 [N] (b*P+pixel, view-major), 'plan': SelectionPlan}

    # -- mining ------------------------------------------------------------------------------------------
    def _mine(self, feats, labels, predict, seg, seg_ready=None, gather=False):
        """Runs cseg_classify_partition. With `seg_ready` (a HIP event recorded right after the logits were produced,
        see nets/hrnet.py) the kernels and the counts D2H copy go to a side stream, so they -- and the host-side
        selection plan that follows -- overlap the projection head still running on the compute stream."""
        B, Dm, h, w = feats.shape

        def run():
            if seg is not None:
                return K.classify_partition(labels, self.ignore_label, seg=seg)
            return K.classify_partition(labels, self.ignore_label, predict=predict.contiguous(),
                                        num_classes=self.configer.get('data', 'num_classes'), feat_hw=(h, w))
        if seg_ready is None or seg is None or not seg.is_cuda:
            return run()
        if self._side is None:
            self._side = torch.cuda.Stream(device=seg.device)
        main = torch.cuda.current_stream(seg.device)
        with torch.cuda.stream(self._side):
            self._side.wait_event(seg_ready)
            cp = run()
            flat = torch.cat([cp["counts"].reshape(-1), cp["status"]])
            if gather:
                # cross-rank contrast set: the all-gather of the per-rank counts is issued HERE, on the side stream, so that it -- like
                # the mining itself and the D2H copy -- runs under the CE kernels / projection head on the compute stream and the host
                # only waits on the event below (round 3: a blocking all-gather + .cpu() in the middle of the loss forward)
                flat = D.all_gather_cat(flat.unsqueeze(0))
            host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
            host.copy_(flat, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in cp.values():
            t.record_stream(main)                # produced on the side stream, consumed on the compute stream
        seg.record_stream(self._side)
        labels.record_stream(self._side)
        main.wait_event(done)
        cp["host_counts_global" if gather else "host_counts"] = (host, done)
        return cp

    def _plan(self, counts, budget_mult=1, draw_images=None):
        plan = plan_selection(counts, self.max_samples * budget_mult, self.max_views, draw_images)
        if plan is None:
            # the reference returns (None, None) and then fails on None.shape (loss_contrast.py:44-45, :92)
            raise RuntimeError("PixelContrastLoss: no (image, class) segment has more than max_views=%d pixels"
                               % self.max_views)
        return plan

    # -- forward -----------------------------------------------------------------------------------------
    def forward(self, feats, labels=None, predict=None, seg=None, seg_ready=None):
        """feats [B,D,h,w] (L2-normalised embeddings), labels [B,H,W] long, and either `predict` [B,h,w] long
        (reference signature, loss_contrast.py:130) or `seg` [B,K,h,w] logits (argmax fused into the mining
        kernel)."""
        assert labels is not None and (predict is not None or seg is not None)
        B, Dm, h, w = feats.shape
        P = h * w
        world = D.get_world_size()
        cross = (world > 1 or D.exercise_single_rank()) and self.cross_rank
        cp = self._mine(feats, labels, predict, seg, seg_ready, gather=cross)
        if cross:
            return self._forward_cross_rank(feats, cp, P, world)
        plan = self._plan(_counts_to_host(cp))
        dev = feats.device
        sel_pos = torch.from_numpy(plan.row_img.astype(np.int32) * P + plan.row_off).to(dev, non_blocking=True)
        a_lab = torch.from_numpy(plan.row_lab.astype(np.int32)).to(dev, non_blocking=True)
        loss, sel_pix = K.PixelContrast.apply(feats, cp["part_idx"], sel_pos, a_lab, "self", self.temperature,
                                              self.base_temperature, None, None, _grad_slot(feats))
        self.last_selection = {"sel_pix": sel_pix, "plan": plan}
        return loss

    def _forward_cross_rank(self, feats, cp, P, world):
        rank = D.get_rank()
        B = feats.shape[0]
        dev = feats.device
        if "host_counts_global" in cp:
            host, done = cp["host_counts_global"]               # gathered and copied on the side stream (_mine)
            done.synchronize()
        else:
            local = torch.cat([cp["counts"].reshape(-1), cp["status"]])
            host = D.all_gather_cat(local.unsqueeze(0)).cpu()  # RCCL all-gather, B*K*2+4 ints per rank
        if int(host[:, -4].sum()) != 0:
            raise RuntimeError("PixelContrastLoss: labels outside [0, num_classes) on some rank")
        counts = host[:, :-4].reshape((world * B,) + tuple(cp["counts"].shape[1:])).numpy()
        # T, n_view, labels and row order are identical on every rank (same counts); the drawn offsets are needed
        # for the rank's own images only
        plan = self._plan(counts, world if self.cross_rank_budget == 'per_rank' else 1,
                          (rank * B, (rank + 1) * B) if self.cross_rank_rng == 'local' else None)
        T, V = plan.T, plan.n_view
        owner = plan.seg_img // B                               # rank of every segment
        mine = np.nonzero(owner == rank)[0]
        t_r = [int((owner == r).sum()) for r in range(world)]
        # local rows, view-major inside this rank's segment range
        rows_local = (np.arange(V)[:, None] * T + mine[None, :]).reshape(-1)
        sel_pos = torch.from_numpy(((plan.row_img[rows_local] - rank * B) * P + plan.row_off[rows_local])
                                   .astype(np.int32)).to(dev)
        anchors_l, sel_pix = K.GatherAnchors.apply(feats, cp["part_idx"], sel_pos, _grad_slot(feats))
        t_max = max(t_r)
        pad = torch.zeros(t_max * V, feats.shape[1], dtype=feats.dtype, device=dev)
        pad[:anchors_l.shape[0]] = anchors_l.detach()
        bufs = D.all_gather_cat(pad.unsqueeze(0))               # RCCL all-gather, <= max_samples x D floats in total
        pieces = []
        order = np.empty(T * V, dtype=np.int64)                 # global row -> position in cat(pieces)
        base = 0
        for r in range(world):
            n_r = t_r[r] * V
            pieces.append(_GradScale.apply(anchors_l, float(world)) if r == rank else bufs[r][:n_r])
            seg_r = np.nonzero(owner == r)[0]
            glob_rows = (np.arange(V)[:, None] * T + seg_r[None, :]).reshape(-1)
            order[glob_rows] = base + np.arange(n_r)
            base += n_r
        allrows = torch.cat(pieces, dim=0)
        anchors_g = allrows.index_select(0, torch.from_numpy(order).to(dev))
        a_lab = torch.from_numpy(plan.row_lab.astype(np.int32)).to(dev)
        loss = K.ContrastOnAnchors.apply(anchors_g, a_lab, "self", self.temperature, self.base_temperature,
                                         None, None, None, None)
        self.last_selection = {"sel_pix": sel_pix, "plan": plan, "rows_local": rows_local}
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
        if self.use_rmi:
            raise NotImplementedError("contrast.use_rmi: the RMI criterion is outside the accelerated hot path")
        self.seg_criterion = FSCELoss(configer=configer)
        self.contrast_criterion = PixelContrastLoss(configer=configer)

    def forward(self, preds, target, with_embed=False):
        assert "seg" in preds
        assert "embed" in preds
        seg = preds['seg']
        embedding = preds['embed']
        loss = self.seg_criterion(seg, target)                       # upsample fused into the CE kernel
        loss_contrast = self.contrast_criterion(embedding, target, seg=seg, seg_ready=preds.get('seg_ready'))
        # the two terms of the last call, detached (no host sync): the segmentation term is a smooth function of the weights, the
        # contrastive term is not (argmax decides hard / easy, rounding-level changes of the logits move anchors between the sets) --
        # tests that compare two implementations after an SGD step bound the former tightly and the latter loosely
        self.last_terms = (loss.detach(), loss_contrast.detach())
        if with_embed is True:
            return loss + self.loss_weight * loss_contrast
        return loss + 0 * loss_contrast  # same trick as the reference: keeps every parameter in the DDP graph


class ContrastAuxCELoss(nn.Module, ABC):
    def __init__(self, configer=None):
        super(ContrastAuxCELoss, self).__init__()
        self.configer = configer
        ignore_index = -1
        if self.configer.exists('loss', 'params') and 'ce_ignore_index' in self.configer.get('loss', 'params'):
            ignore_index = self.configer.get('loss', 'params')['ce_ignore_index']
        Log.info('ignore_index: {}'.format(ignore_index))
        self.loss_weight = self.configer.get('contrast', 'loss_weight')
        self.use_rmi = self.configer.get('contrast', 'use_rmi')
        if self.use_rmi:
            raise NotImplementedError("contrast.use_rmi: the RMI criterion is outside the accelerated hot path")
        self.seg_criterion = FSAuxCELoss(configer=configer)
        self.contrast_criterion = PixelContrastLoss(configer=configer)

    def forward(self, preds, target, with_embed=False):
        assert "seg" in preds
        assert "seg_aux" in preds
        assert "embed" in preds
        seg = preds['seg']
        seg_aux = preds['seg_aux']
        embedding = preds['embed']
        loss = self.seg_criterion([seg_aux, seg], target)
        loss_contrast = self.contrast_criterion(embedding, target, seg=seg, seg_ready=preds.get('seg_ready'))
        # the two terms of the last call, detached (no host sync): the segmentation term is a smooth function of the weights, the
        # contrastive term is not (argmax decides hard / easy, rounding-level changes of the logits move anchors between the sets) --
        # tests that compare two implementations after an SGD step bound the former tightly and the latter loosely
        self.last_terms = (loss.detach(), loss_contrast.detach())
        if with_embed is True:
            return loss + self.loss_weight * loss_contrast
        return loss + 0 * loss_contrast
