"""Pixel-embedding projection head (reference: lib/models/modules/projection.py:8-24).

`proj='convmlp'`: 1x1 conv (dim_in -> dim_in) -> BN + ReLU -> 1x1 conv (dim_in -> proj_dim); `proj='linear'`: a single
1x1 conv. The output is L2-normalised over channels (eps 1e-12), which is what makes anchor . contrast a cosine
similarity in the contrastive kernels. Parameter names (`proj.0`, `proj.1.0`, `proj.2` / `proj`) and creation order
match the reference so checkpoints and seeds interchange; the 1x1 convolutions are GEMMs on MIOpen/rocBLAS.

Row-sparse backward (opt-in: kernels.SPARSE_EMBED_GRAD / env CSEG_SPARSE_EMBED_GRAD=1, first hardware run pending).
The contrastive term reads <= max_samples pixels of the embedding (loss_contrast.py:66-87 of the reference), so the
gradient that comes back is N <= 1024 rows of [B*h*w] = 262 144 at the benched shape. The dense route pays for that
structure being invisible to autograd: a 268 MB zero tensor + scatter, the dense adjoint of F.normalize (6 passes over
268 MB), the dense 256 -> 720 backward-data and weight-gradient GEMMs of `proj.2` (2 x 97 GFLOP) and a dense BN
backward over 755 MB. `_SparseTail` spans BN+ReLU -> proj.2 -> normalise as one autograd node: the forward is the same
three device ops; the backward receives the rows through a kernels.SparseGradSlot (the loss' backward deposits them
and returns a storage-free zero stand-in), recomputes the activations of those N pixels from the saved conv output,
does the normalise / 1x1 / ReLU adjoints on [N, C] matrices, and because the BN statistics gradient sums are sums over
N pixels, the dense part of d(conv output) collapses to ONE per-channel affine map of the saved input
(du = A_c + B_c * u, one read + one write) plus N corrected rows. The node saves neither the BN output (755 MB) nor the
un-normalised embedding (268 MB). If the embedding has any other consumer, autograd hands over a real dense gradient
instead of the stand-in and the node falls back to the dense adjoint (recomputing what it did not save)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from contrastiveseg_amd import kernels as K          # device ops (CPU tests inject oracle/cpu_port.py here)
from contrastiveseg_amd import kernels as _switches  # SPARSE_EMBED_GRAD / SparseGradSlot: host-side, never injected
from contrastiveseg_amd.lib.models.tools import fused_bn
from contrastiveseg_amd.lib.models.tools.module_helper import Conv1x1, ModuleHelper
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

_KINDS = ('linear', 'convmlp')
_EPS = 1e-12           # F.normalize's default, what the reference uses


def _pointwise(cin, cout):
    return Conv1x1(cin, cout)            # an nn.Conv2d(cin, cout, 1); split-bf16 kernel when switched on


def _split_pixels(sel_pix, P):
    sel = sel_pix.long()
    return torch.div(sel, P, rounding_mode='floor'), sel % P


def _add_rows(t, b_idx, p_idx, rows):
    """t[b, :, p] += rows[n, :] for the N (image, pixel) pairs, in place on the CONTIGUOUS [B, C, P] tensor. The pairs are
    duplicate-free, so the atomic adds of index_add_ never meet and the result is deterministic. (An index_put_ through the
    permuted view [B, P, C] made PyTorch copy the whole 755 MB tensor into that layout and back: 1.0 + 0.8 ms per step in
    the round-3 trace.)"""
    B, C, P = t.shape[0], t.shape[1], t.shape[2] * t.shape[3] if t.dim() == 4 else t.shape[2]
    lin = ((b_idx * C) * P + p_idx).unsqueeze(1) + torch.arange(C, device=t.device, dtype=torch.long).unsqueeze(0) * P
    t.view(-1).index_add_(0, lin.reshape(-1), rows.reshape(-1).to(t.dtype))
    return t


class _SparseTail(torch.autograd.Function):
    """e = normalize(conv1x1(relu(bn(u)), w2, b2)); see the module docstring for the backward."""

    @staticmethod
    def forward(ctx, u, gamma, beta, w2, b2, bn, slot):
        u = u.contiguous()
        training = bn.training or not bn.track_running_stats
        track = bn.track_running_stats
        group = bn._sync_group() if training else None
        a, mi, count = fused_bn.bn_forward(
            u, gamma, beta, None, bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if (track and bn.training) else None, training, True, float(bn.momentum),
            float(bn.eps), group)
        # proj.2 (720 -> 256 at 1/4 resolution: 97 GFLOP at batch 8) on the split-operand 1x1 kernel where it applies (rocBLAS fp32:
        # 0.59 ms in the round-4 trace), else MIOpen / rocBLAS
        if (_switches.CONV1X1_SPLIT_BF16 and _switches._on_device(a) and _switches.conv1x1_sb_eligible(a, w2)
                and _switches.conv1x1_sb_tiles(a, w2.shape[0]) >= _switches.CONV1X1_SB_MIN_TILES):
            z = K.conv1x1_sb_run(a, w2, False, b2, ax=_switches.known_amax(a))
        else:
            z = F.conv2d(a, w2, b2)
        e = F.normalize(z, p=2, dim=1, eps=_EPS)
        ctx.save_for_backward(u, mi, gamma, beta, w2, b2)
        ctx.meta = (training, count, group)
        ctx.slot = slot
        return e

    @staticmethod
    def backward(ctx, g):
        u, mi, gamma, beta, w2, b2 = ctx.saved_tensors
        training, count, group = ctx.meta
        slot = ctx.slot
        standin = slot.is_standin(g)             # before take(): that drops the stand-in
        deposits = slot.take()
        B, C, h, w = u.shape
        P, D = h * w, w2.shape[0]
        want_du = ctx.needs_input_grad[0]
        if deposits:
            rows = deposits[0][0] if len(deposits) == 1 else torch.cat([d[0] for d in deposits], dim=0)
            sel = deposits[0][1] if len(deposits) == 1 else torch.cat([d[1] for d in deposits], dim=0)
            b_idx, p_idx = _split_pixels(sel, P)
        if not (standin and deposits):
            # dense route: some other consumer contributed a real gradient (or nothing was deposited)
            gt = g.contiguous()
            if deposits:
                gt = _add_rows(gt.clone(), b_idx, p_idx, rows)
            a = K.bn_apply(u, mi, gamma, beta, None, True)
            with torch.enable_grad():
                a_ = a.detach().requires_grad_(True)
                w_ = w2.detach().requires_grad_(True)
                b_ = None if b2 is None else b2.detach().requires_grad_(True)
                e = F.normalize(F.conv2d(a_, w_, b_), p=2, dim=1, eps=_EPS)
                wrt = (a_, w_) + (() if b_ is None else (b_,))
                got = torch.autograd.grad(e, wrt, gt)
            d_a, d_w2, d_b2 = got[0], got[1], (got[2] if b_ is not None else None)
            du, d_gamma, d_beta, _ = fused_bn.bn_backward(d_a.contiguous(), u, None, mi, gamma, beta, True, False,
                                                         training, count, group, want_du)
            return (du, d_gamma if gamma is not None else None, d_beta if beta is not None else None, d_w2, d_b2,
                    None, None)
        # ---- row-sparse route: everything below is [N, C] / [N, D] / per-channel, except the one affine pass over u
        rows = rows.to(u.dtype)
        mean, invstd = mi[:, 0], mi[:, 1]
        scale = invstd if gamma is None else invstd * gamma           # d(pre)/d(u) per channel
        uf = u.view(B, C, P)
        xhat = (uf[b_idx, :, p_idx] - mean) * invstd                  # [N, C]
        pre = xhat if gamma is None else xhat * gamma
        if beta is not None:
            pre = pre + beta
        a_sel = pre.clamp_min(0)
        W = w2.view(D, C)
        z = a_sel @ W.t()
        if b2 is not None:
            z = z + b2
        nrm = z.norm(dim=1, keepdim=True)
        den = nrm.clamp_min(_EPS)
        e = z / den
        # adjoint of z / max(|z|, eps): the norm term vanishes where the clamp is active
        d_z = torch.where(nrm > _EPS, rows - e * (e * rows).sum(1, keepdim=True), rows) / den
        d_w2 = (d_z.t() @ a_sel).view_as(w2)
        d_b2 = d_z.sum(0) if b2 is not None else None
        d_pre = (d_z @ W) * (pre > 0).to(d_z.dtype)
        sums = torch.stack([d_pre.double().sum(0), (d_pre * xhat).double().sum(0)], dim=1)   # [C, 2]: rank-local
        d_beta = sums[:, 0].to(u.dtype) if beta is not None else None
        d_gamma = sums[:, 1].to(u.dtype) if gamma is not None else None
        du = None
        if want_du:
            vals = d_pre * scale
            if training:
                if group is not None:
                    sums = fused_bn._all_reduce(sums.clone(), group)
                # du = scale * (d_pre - m1 - xhat * m2), with d_pre zero outside the N pixels:
                #    = A_c + B_c * u  everywhere   (+ scale * d_pre at the N pixels); coefficients in fp64
                m = sums / count
                s64, i64, mu64 = scale.double(), invstd.double(), mean.double()
                b_c = -(s64 * i64 * m[:, 1])
                a_c = -(s64 * m[:, 0]) - b_c * mu64
                am = None
                if (_switches.PRODUCER_AMAX and _switches._on_device(u) and u.dtype == torch.float32 and P % 4 == 0
                        and hasattr(K, 'affine_channels')):
                    # one kernel: the affine map AND the max|du| record for proj.0's backward-data / weight-gradient kernels
                    # (torch.addcmul + a max|.| pass over the 755 MB tensor before)
                    du, am = K.affine_channels(u, a_c.to(u.dtype), b_c.to(u.dtype), _switches.split_arith_id() != 0)
                else:
                    du = torch.addcmul(a_c.to(u.dtype).view(1, C, 1, 1), u, b_c.to(u.dtype).view(1, C, 1, 1))
            else:
                du, am = torch.zeros_like(u), None            # frozen statistics: only the N pixels carry gradient
            _add_rows(du, b_idx, p_idx, vals)
            if am is not None:
                # the N corrected rows may exceed the affine part: their final values join the record (a [N, C] gather)
                _switches.tensor_amax(du.view(B, C, P)[b_idx, :, p_idx].contiguous(), slot=am)
                _switches.amax_attach(du, am)
            if training:
                _switches.mark_zero_channel_sum(du)           # batch statistics: sum(du) = 0 per channel (kernels.bias_grad)
        return du, d_gamma, d_beta, d_w2, d_b2, None, None


class ProjectionHead(nn.Module):
    def __init__(self, dim_in, proj_dim=256, proj='convmlp', bn_type='torchsyncbn'):
        super(ProjectionHead, self).__init__()
        if proj not in _KINDS:
            raise ValueError('unknown projection {!r}; expected one of {}'.format(proj, _KINDS))
        Log.info('proj_dim: {}'.format(proj_dim))
        self.dim_in, self.proj_dim, self.kind = dim_in, proj_dim, proj
        if proj == 'linear':
            self.proj = _pointwise(dim_in, proj_dim)
        else:
            stages = [_pointwise(dim_in, dim_in)]
            stages.append(ModuleHelper.BNReLU(dim_in, bn_type=bn_type))
            stages.append(_pointwise(dim_in, proj_dim))
            self.proj = nn.Sequential(*stages)

    def extra_repr(self):
        return 'dim_in={}, proj_dim={}, kind={}'.format(self.dim_in, self.proj_dim, self.kind)

    def forward(self, x):
        if self.kind == 'convmlp' and _switches.SPARSE_EMBED_GRAD and torch.is_grad_enabled() and x.requires_grad:
            bn, last = self.proj[1][0], self.proj[2]
            if isinstance(bn, fused_bn._FusedMixin) and bn.momentum is not None:
                slot = _switches.SparseGradSlot()
                e = _SparseTail.apply(self.proj[0](x), bn.weight, bn.bias, last.weight, last.bias, bn, slot)
                e._cseg_grad_slot = slot          # read by the contrastive criteria (lib/loss/loss_contrast*.py)
                return e
        return F.normalize(self.proj(x), p=2, dim=1)
