"""
ORACLE -- test infrastructure only.

Torch-CPU (fp32, autograd) restatement of the device half of the hot path with the same call surface as
`contrastiveseg_amd.kernels`, so that
  * the host logic of the product (loss modules, trainer, cross-rank exchange) can be exercised on CPU in
    `-m "not gpu"` tests by injecting this module in place of the HIP binding (tests only: the product has no CPU
    path and fails loudly without libcseg_hip.so), and
  * bench.py's `cpu_baseline` leg can time a CPU port of the train step on the GPU box's host cores.
Each function restates the reference lines cited in oracle/cseg_oracle.py; results are pinned against the numpy
oracle (and through it against the reference-generated golden vectors) in tests/test_cpu_port.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---- mining ------------------------------------------------------------------------------------------------
@torch.no_grad()
def classify_partition(target, ignore_label, seg=None, predict=None, num_classes=None, feat_hw=None, want_maps=False):
    B, H, W = target.shape
    if seg is not None:
        K, h, w = seg.shape[1:]
        pred = torch.max(seg, 1)[1].reshape(B, -1)                       # loss_contrast.py:183
    else:
        K = int(num_classes)
        h, w = feat_hw
        pred = predict.reshape(B, -1)
    lab = F.interpolate(target.unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1).long().reshape(B, -1)  # :131-134
    P = h * w
    counts = torch.zeros(B, K, 2, dtype=torch.int32)
    seg_off = torch.zeros(B, K, 2, dtype=torch.int32)
    part = torch.zeros(B, P, dtype=torch.int32)
    bad = int(((lab != ignore_label) & ((lab < 0) | (lab >= K))).sum())
    for b in range(B):
        off = 0
        for c in range(K):
            hard = ((lab[b] == c) & (pred[b] != c)).nonzero().reshape(-1)   # :60
            easy = ((lab[b] == c) & (pred[b] == c)).nonzero().reshape(-1)   # :61
            counts[b, c, 0], counts[b, c, 1] = len(hard), len(easy)
            seg_off[b, c, 0] = off
            part[b, off:off + len(hard)] = hard.int()
            off += len(hard)
            seg_off[b, c, 1] = off
            part[b, off:off + len(easy)] = easy.int()
            off += len(easy)
    out = {"counts": counts, "seg_off": seg_off, "part_idx": part,
           "status": torch.tensor([bad, 0, 0, 0], dtype=torch.int32)}
    if want_maps:
        out["lab"], out["pred"] = lab.int(), pred.int()
    return out


def _rows(embed, part_idx, sel_pos):
    B, D = embed.shape[:2]
    P = embed.shape[2] * embed.shape[3]
    pos = sel_pos.long()
    b = pos // P
    pix = part_idx.reshape(-1)[pos].long()
    flat = embed.reshape(B, D, P)
    return flat[b, :, pix], (b * P + pix).int()          # [N, D]


def _contrast(A, ya, C, yc, temperature, base_temperature):
    """loss_contrast.py:91-128 / loss_contrast_mem.py:107-152 on already view-major rows."""
    N = A.shape[0]
    adc = torch.matmul(A, C.T) / temperature
    logits = adc - adc.max(dim=1, keepdim=True)[0].detach()
    same = (ya.reshape(-1, 1) == yc.reshape(1, -1)).float()
    neg_mask = 1 - same
    logits_mask = torch.ones_like(same).scatter_(1, torch.arange(N).reshape(-1, 1), 0)
    mask = same * logits_mask
    neg = (torch.exp(logits) * neg_mask).sum(1, keepdim=True)
    log_prob = logits - torch.log(torch.exp(logits) + neg)
    mlpp = (mask * log_prob).sum(1) / mask.sum(1)
    return (-(temperature / base_temperature) * mlpp).mean()


def _bank(segment_queue, pixel_queue):
    """loss_contrast_mem.py:91-105 on cat(segment_queue, pixel_queue, dim=1)."""
    Q = torch.cat((segment_queue, pixel_queue), dim=1)
    Kc, S, D = Q.shape
    X = torch.zeros(Kc * S, D)
    y = torch.zeros(Kc * S)
    X[:(Kc - 1) * S] = Q[1:].reshape(-1, D)
    y[:(Kc - 1) * S] = torch.arange(1, Kc).repeat_interleave(S).float()
    return X, y


class _Apply(object):
    def __init__(self, fn):
        self.apply = fn


class _RowsDeposit(torch.autograd.Function):
    """_rows whose backward hands the row gradient to a kernels.SparseGradSlot (the product's PixelContrast /
    GatherAnchors do the same when the projection head attached a slot to the embedding)."""

    @staticmethod
    def forward(ctx, embed, part_idx, sel_pos, slot):
        A, sel_pix = _rows(embed, part_idx, sel_pos)
        ctx.slot, ctx.sel_pix, ctx.shape = slot, sel_pix, embed.shape
        ctx.mark_non_differentiable(sel_pix)
        return A, sel_pix

    @staticmethod
    def backward(ctx, g, _):
        return ctx.slot.deposit(g.contiguous(), ctx.sel_pix, ctx.shape), None, None, None


def _gather(embed, part_idx, sel_pos, slot=None):
    if slot is None:
        return _rows(embed, part_idx, sel_pos)
    return _RowsDeposit.apply(embed, part_idx, sel_pos, slot)


def _pixel_contrast(embed, part_idx, sel_pos, a_lab, mode, temperature, base_temperature, segment_queue, pixel_queue,
                    slot=None):
    A, sel_pix = _gather(embed, part_idx, sel_pos, slot)
    if mode == "self":
        loss = _contrast(A, a_lab, A, a_lab, temperature, base_temperature)
    else:
        C, yc = _bank(segment_queue, pixel_queue)
        loss = _contrast(A, a_lab.float(), C, yc, temperature, base_temperature)
    return loss, sel_pix


def _contrast_on_anchors(anchors, a_lab, mode, temperature, base_temperature, contrast, c_lab, segment_queue,
                         pixel_queue):
    if mode == "self":
        return _contrast(anchors, a_lab, anchors, a_lab, temperature, base_temperature)
    if mode == "plain":
        return _contrast(anchors, a_lab, contrast, c_lab, temperature, base_temperature)
    C, yc = _bank(segment_queue, pixel_queue)
    return _contrast(anchors, a_lab.float(), C, yc, temperature, base_temperature)


PixelContrast = _Apply(_pixel_contrast)
ContrastOnAnchors = _Apply(_contrast_on_anchors)
GatherAnchors = _Apply(_gather)


# ---- head / CE ----------------------------------------------------------------------------------------------
def upsample_concat(feats):
    h, w = feats[0].shape[-2:]                                               # nets/hrnet.py:86-91
    return torch.cat([feats[0]] + [F.interpolate(f, size=(h, w), mode="bilinear", align_corners=True)
                                   for f in feats[1:]], 1)


def fuse_sum_relu(same, low):
    """HighResolutionModule.forward exchange step (hrnet_backbone.py:271-286): branch-ordered sum, then ReLU."""
    y = same[0]
    for t in same[1:]:
        y = y + t
    for t in low:
        y = y + F.interpolate(t, size=y.shape[-2:], mode="bilinear", align_corners=True)
    return F.relu(y)


FANOUT_SUM = False          # kernels.FANOUT_SUM: the gradient sums of the exchange unit are autograd's own here


def fan_out(x, n):
    """kernels.fan_out: n consumers of one tensor (autograd accumulates their gradients)."""
    return [x] * n


def upsample_ce(seg, target, weight=None, ignore_index=-1, status=None):
    pred = F.interpolate(seg, size=target.shape[-2:], mode="bilinear", align_corners=True)   # loss_contrast.py:180
    return F.cross_entropy(pred, target, weight=weight, ignore_index=ignore_index)            # loss_helper.py:186


# ---- memory bank ----------------------------------------------------------------------------------------------
@torch.no_grad()
def queue_count(labels, stride, num_classes):
    lab = labels[:, ::stride, ::stride].reshape(labels.shape[0], -1)
    out = torch.zeros(labels.shape[0], num_classes, dtype=torch.int32)
    for b in range(lab.shape[0]):
        v = lab[b][(lab[b] >= 0) & (lab[b] < num_classes)]
        out[b] = torch.bincount(v, minlength=num_classes).int()
    return out


@torch.no_grad()
def queue_class_sums(keys, labels, stride, num_classes):
    B, D = keys.shape[:2]
    lab = labels[:, ::stride, ::stride].reshape(B, -1)
    feat = keys.reshape(B, D, -1)
    sums = torch.zeros(B, num_classes, D)
    for b in range(B):
        for c in range(num_classes):
            idx = (lab[b] == c).nonzero().reshape(-1)
            if len(idx):
                sums[b, c] = feat[b][:, idx].sum(1)
    return sums


@torch.no_grad()
def queue_write_segments(sums, counts, job_img, job_cls, job_dst_row, segment_queue):
    for b, c, r in zip(job_img.tolist(), job_cls.tolist(), job_dst_row.tolist()):
        segment_queue[c, r] = F.normalize(sums[b, c] / counts[b, c].float(), p=2, dim=0)


@torch.no_grad()
def queue_write_pixels(keys, src_img, src_pos, dst_cls, dst_row, pixel_queue):
    feat = keys.reshape(keys.shape[0], keys.shape[1], -1)
    for b, p, c, r in zip(src_img.tolist(), src_pos.tolist(), dst_cls.tolist(), dst_row.tolist()):
        pixel_queue[c, r] = F.normalize(feat[b, :, p], p=2, dim=0)


class wgrad_scope(object):
    """kernels.wgrad_scope on the CPU: nothing to fork."""

    def __init__(self, device):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def known_tile_stats(t):
    """The torch restatement has no convolution epilogue: BatchNorm always takes its own statistics pass."""
    return None


# ---- fused (Sync)BatchNorm + residual + ReLU primitives (csrc/bn.hip), torch restatement ------------------------
# What they restate: nn.BatchNorm2d / nn.SyncBatchNorm training forward+backward (module_helper.py:29-68 of the
# reference -> torch's batch_norm) followed by the add / ReLU of the residual blocks. Statistics in fp64 like the kernel.
def _bn_flat(x):
    return x.reshape(x.shape[0], x.shape[1], -1)


@torch.no_grad()
def bn_stats(x):
    """[C+1,2]: row C = (this rank's element count per channel, 0), as cseg_bn_stats (ABI 4)."""
    f = _bn_flat(x).double()
    m = torch.stack([f.sum((0, 2)), (f * f).sum((0, 2))], dim=1)
    n = torch.tensor([[float(f.shape[0] * f.shape[2]), 0.0]], dtype=torch.float64, device=x.device)
    return torch.cat([m, n], dim=0)


@torch.no_grad()
def bn_finalize(moments, count, eps, momentum, running_mean, running_var, num_batches_tracked):
    if count == 0:
        count = float(moments[-1, 0])              # the exchanged count (the kernel reads it on the device)
    moments = moments[:-1]
    mean = moments[:, 0] / count
    var = (moments[:, 1] / count - mean * mean).clamp_(min=0.0)
    if running_mean is not None:
        unbiased = var * (count / (count - 1.0)) if count > 1 else var
        running_mean.copy_(((1.0 - momentum) * running_mean.double() + momentum * mean).float())
        running_var.copy_(((1.0 - momentum) * running_var.double() + momentum * unbiased).float())
    if num_batches_tracked is not None:
        num_batches_tracked += 1
    return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=1).float()


@torch.no_grad()
def bn_stats_finalize(x, eps, momentum, running_mean, running_var, num_batches_tracked):
    return bn_finalize(bn_stats(x), float(x.numel() // x.shape[1]), eps, momentum, running_mean, running_var,
                       num_batches_tracked)


def _bn_affine(x, mean_invstd, weight, bias):
    shape = (1, -1) + (1,) * (x.dim() - 2)
    a = mean_invstd[:, 1] if weight is None else weight * mean_invstd[:, 1]
    xm = x - mean_invstd[:, 0].reshape(shape)
    z = xm * a.reshape(shape)
    if bias is not None:
        z = z + bias.reshape(shape)
    return xm, a, z


@torch.no_grad()
def bn_apply(x, mean_invstd, weight, bias, residual, relu, amax=None):
    _, _, z = _bn_affine(x, mean_invstd, weight, bias)
    if residual is not None:
        z = z + residual
    return F.relu(z) if relu else z


@torch.no_grad()
def bn_bwd_reduce(dy, x, out, mean_invstd, weight, bias, mode):
    xm, _, z = _bn_affine(x, mean_invstd, weight, bias)
    g = dy
    if mode == 1:
        g = dy * (z > 0)
    elif mode == 2:
        g = dy * (out > 0)
    s0 = _bn_flat(g).double().sum((0, 2))
    s1 = (_bn_flat(g).double() * _bn_flat(xm).double()).sum((0, 2))
    n = torch.tensor([[float(x.numel() // x.shape[1]), 0.0]], dtype=torch.float64, device=x.device)
    sums = torch.cat([torch.stack([s0, s1], dim=1), n], dim=0)
    return sums, (s1 * mean_invstd[:, 1].double()).float(), s0.float(), (g if mode == 2 else None)


@torch.no_grad()
def bn_fwd(x, weight, bias, residual, relu, eps, momentum, running_mean, running_var, num_batches_tracked, amax=None):
    mi = bn_stats_finalize(x, eps, momentum, running_mean, running_var, num_batches_tracked)
    return bn_apply(x, mi, weight, bias, residual, relu), mi


@torch.no_grad()
def bn_bwd(dy, x, out, mean_invstd, weight, bias, mode, training, want_dx, amax=None):
    sums, d_weight, d_bias, g = bn_bwd_reduce(dy, x, out, mean_invstd, weight, bias, mode)
    dx = None
    if want_dx:
        dx = bn_bwd_apply(g if mode == 2 else dy, x, mean_invstd, weight, bias, sums if training else None,
                          float(x.numel() // x.shape[1]), mode == 1)
    return dx, d_weight, d_bias, g


@torch.no_grad()
def bn_bwd_apply(dy, x, mean_invstd, weight, bias, sums, count, mask_from_x, amax=None):
    xm, a, z = _bn_affine(x, mean_invstd, weight, bias)
    shape = (1, -1) + (1,) * (x.dim() - 2)
    g = dy * (z > 0) if mask_from_x else dy
    if sums is None:
        return g * a.reshape(shape)
    if count == 0:
        count = float(sums[-1, 0])
    sums = sums[:-1]
    k0d = sums[:, 0] / count
    k0 = k0d.float()
    k0l = (k0d - k0.double()).float().reshape(shape)      # mean(dy') carried as hi + lo, like the kernel
    k0 = k0.reshape(shape)
    k1 = (sums[:, 1] / count * mean_invstd[:, 1].double() ** 2).float().reshape(shape)
    return a.reshape(shape) * ((g - k0) - k0l - xm * k1)


# ---- GPU data pipeline (csrc/augment.hip), restated through the forward chain of oracle/aug_oracle.py -------------
@torch.no_grad()
def augment_batch(img_u8, lab_u8, lut, params, out_hw, div_value, mean, std):
    from oracle import aug_oracle
    Ht, Wt = out_hw
    imgs, labs = [], []
    for b in range(img_u8.shape[0]):
        im, lb = aug_oracle.apply_chain(img_u8[b].numpy(), None if lab_u8 is None else lab_u8[b].numpy(),
                                        params[b].numpy(), (Wt, Ht), div_value, mean, std,
                                        None if lut is None else lut.numpy())
        imgs.append(torch.from_numpy(im))
        labs.append(None if lb is None else torch.from_numpy(lb))
    return torch.stack(imgs), (None if lab_u8 is None else torch.stack(labs))


def amax_request(t):
    """kernels.amax_request: the CPU restatement has no split-operand convolutions, so nobody asks for max|t|."""
    return None


def amax_attach(t, slot):
    return t


def install(monkeypatch_or_none=None):
    """Points the product's loss / model / trainer modules at this CPU restatement. TESTS AND THE cpu_baseline LEG
    ONLY. Returns a function that restores the HIP binding."""
    import sys
    me = sys.modules[__name__]
    import contrastiveseg_amd.lib.loss.loss_contrast as lc
    import contrastiveseg_amd.lib.loss.loss_contrast_mem as lm
    import contrastiveseg_amd.lib.loss.loss_helper as lh
    import contrastiveseg_amd.lib.models.backbones.hrnet_backbone as hb
    import contrastiveseg_amd.lib.models.nets.hrnet as nh
    import contrastiveseg_amd.lib.models.tools.fused_bn as fb
    import contrastiveseg_amd.lib.models.modules.projection as pj
    import contrastiveseg_amd.lib.datasets.tools.gpu_aug as ga
    import contrastiveseg_amd.segmentor.trainer_contrastive as tc
    mods = [lc, lm, lh, nh, hb, tc, fb, ga, pj]
    saved = [m.K for m in mods]
    for m in mods:
        if monkeypatch_or_none is not None:
            monkeypatch_or_none.setattr(m, "K", me)
        else:
            m.K = me

    def restore():
        for m, k in zip(mods, saved):
            m.K = k
    return restore
