"""FusedBatchNorm2d / FusedSyncBatchNorm: drop-in subclasses of nn.BatchNorm2d / nn.SyncBatchNorm (same parameters,
buffers and state_dict keys as what ModuleHelper returns in the reference, lib/models/tools/module_helper.py:29-68) whose
forward optionally fuses the residual add and the ReLU that follow every BN of the hot-path networks:

    y = bn(x)                          ->  bn(x)
    y = relu(bn(x))                    ->  bn(x, relu=True)            (BNReLU, conv-bn-relu chains)
    y = relu(bn(x) + residual)         ->  bn(x, residual=r, relu=True) (BasicBlock / Bottleneck tails)

Device work = the cseg_bn_* kernels (csrc/bn.hip) reached through `K` (contrastiveseg_amd.kernels; tests inject the
torch restatement oracle/cpu_port.py). Host logic here: training/eval switch, the SyncBN exchange and autograd wiring.

SyncBN exchange, MI355X-first: ONE all-reduce (RCCL) of the packed per-channel fp64 moments [C,2] in forward and ONE of
the [C,2] gradient sums in backward, instead of torch's all_gather of (mean, invstd, count) + gather-stats kernel and a
separate all-reduce in backward. d_weight / d_bias stay rank-local sums (DDP averages them), exactly like
torch.nn.SyncBatchNorm. With equal per-rank batch sizes the result equals single-process BN on the concatenated batch."""
import torch
import torch.nn as nn

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.utils import distributed as D


def _all_reduce(t, group):
    import torch.distributed as dist
    dist.all_reduce(t, group=group)
    return t


# Element counts under SyncBN. The exchange sums, with the moments, each rank's element count per channel (row C of the
# [C+1,2] fp64 tensor cseg_bn_stats / cseg_bn_bwd_reduce write, ABI 4), and cseg_bn_finalize / cseg_bn_bwd_apply read the
# summed count ON THE DEVICE (count argument 0). Ranks may therefore contribute different batch sizes -- a last partial
# batch, a custom loader -- exactly like torch.nn.SyncBatchNorm (which all_gathers the counts), with no second collective,
# no host round trip and nothing cached per shape. (Round 3 assumed equal batches and verified the assumption with an
# extra MAX all-reduce that was skipped for shapes a rank had seen before: a rank with a new shape issued it while its
# peers did not, and the collectives fell out of step -- ADVICE r3.)
SYNC_COUNT = 0.0          # "take the count from the exchanged tensor"


def bn_forward(x, weight, bias, residual, running_mean, running_var, num_batches_tracked, training, relu, momentum, eps,
               sync_group):
    """Device half of one BN(+residual)(+ReLU) site -> (y, mean_invstd [C,2], count). x / residual contiguous.
    sync_group: None = local statistics; otherwise the process group whose ranks share statistics. count: a float
    (local statistics) or a 0-dim fp64 DEVICE tensor (synchronised: the summed per-rank counts, never read by the host)."""
    n_local = x.numel() // x.shape[1]
    count = float(n_local)
    # max|y|, accumulated by the apply kernel while it stores y: the next split-operand convolution (f16x3 arithmetic) scales
    # its input with it and would otherwise spend a pass over y on it (kernels.amax_of)
    amax = K.amax_request(x)
    # statistics the producing convolution's epilogue already wrote (csrc/cseg_stats.h), or None = one pass over x
    tiles = K.known_tile_stats(x) if training else None
    if training:
        if sync_group is not None:
            moments = _all_reduce(K.bn_tiles_moments(tiles) if tiles is not None else K.bn_stats(x), sync_group)   # [C+1,2]: row C = summed counts
            count = moments[-1, 0]
            mi = K.bn_finalize(moments, SYNC_COUNT, eps, momentum, running_mean, running_var, num_batches_tracked)
            y = K.bn_apply(x, mi, weight, bias, residual, relu, amax=amax)
        elif tiles is not None:
            # single rank, statistics from the convolution's epilogue: a per-channel combine of the segment records + the apply pass
            mi = K.bn_tiles_finalize(tiles, eps, momentum, running_mean, running_var, num_batches_tracked)
            y = K.bn_apply(x, mi, weight, bias, residual, relu, amax=amax)
        else:
            # single rank: statistics + (finalise, running statistics, apply) in two launches
            y, mi = K.bn_fwd(x, weight, bias, residual, relu, eps, momentum, running_mean, running_var,
                             num_batches_tracked, amax=amax)
    else:
        mi = torch.stack([running_mean, torch.rsqrt(running_var + eps)], dim=1).contiguous()
        y = K.bn_apply(x, mi, weight, bias, residual, relu, amax=amax)
    K.amax_attach(y, amax)
    return y, mi, count


def bn_backward(dy, x, out, mi, weight, bias, relu, has_res, training, count, sync_group, want_dx):
    """Adjoint of bn_forward -> (dx or None, d_weight, d_bias, gradient of the residual or None). dy contiguous; `out` is
    the forward's output when a residual was added under the ReLU (the mask cannot be rebuilt from x then)."""
    mode = 0 if not relu else (2 if has_res else 1)
    amax = K.amax_request(x) if want_dx else None      # max|dx|: dx is the output gradient of the convolution in front
    if sync_group is None or not training:
        # single rank (or frozen statistics): reduce + (sums, parameter gradients, dx) in two launches
        dx, d_weight, d_bias, g = K.bn_bwd(dy, x, out, mi, weight, bias, mode, training, want_dx, amax=amax)
    else:
        sums, d_weight, d_bias, g = K.bn_bwd_reduce(dy, x, out, mi, weight, bias, mode)
        dx = None
        if want_dx:
            sums = _all_reduce(sums, sync_group)                      # [C+1,2]: row C = summed element counts again
            dx = K.bn_bwd_apply(g if mode == 2 else dy, x, mi, weight, bias, sums, SYNC_COUNT, mode == 1, amax=amax)
    if dx is not None:
        K.amax_attach(dx, amax)
    d_res = None
    if has_res:
        d_res = g if mode == 2 else dy          # the add passes the (masked) gradient straight through
    return dx, d_weight, d_bias, d_res


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, num_batches_tracked, training, relu,
                momentum, eps, sync_group):
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        y, mi, count = bn_forward(x, weight, bias, residual, running_mean, running_var, num_batches_tracked, training,
                                  relu, momentum, eps, sync_group)
        ctx.meta = (training, relu, residual is not None, count, sync_group)
        ctx.save_for_backward(x, mi, weight, bias, y if (relu and residual is not None) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mi, weight, bias, out = ctx.saved_tensors
        training, relu, has_res, count, sync_group = ctx.meta
        dx, d_weight, d_bias, d_res = bn_backward(dy.contiguous(), x, out, mi, weight, bias, relu, has_res, training,
                                                  count, sync_group, ctx.needs_input_grad[0])
        return (dx, d_weight if weight is not None else None, d_bias if bias is not None else None,
                d_res if ctx.needs_input_grad[3] else None, None, None, None, None, None, None, None, None)


class _BNActGroup(torch.autograd.Function):
    """Several independent SyncBN(+residual)(+ReLU) sites evaluated together so that their statistics travel in ONE
    all-reduce per direction: the moments [sum C_i, 2] (fp64) of all sites are concatenated forward, the gradient sums
    backward. Used for BN sites that sit at the same depth of parallel HRNet branches / exchange paths (they have no data
    dependence on each other), which cuts the SyncBN collectives of an HRNet-W48 step from 2 x 307 to about 2 x 130.
    Training mode with a process group only; the single-rank path keeps the two-launch per-site kernels."""

    @staticmethod
    def forward(ctx, meta, sync_group, *tensors):
        # meta: per site (running_mean, running_var, num_batches_tracked, relu, momentum, eps)
        # tensors: per site x, weight, bias, residual (None allowed for weight / bias / residual)
        n = len(meta)
        xs, moments = [], []
        for i in range(n):
            x = tensors[4 * i].contiguous()
            xs.append(x)
            tiles = K.known_tile_stats(x)                 # the producing convolution's epilogue statistics, if it wrote them
            moments.append(K.bn_tiles_moments(tiles) if tiles is not None else K.bn_stats(x))     # [C_i + 1, 2]: last row = element count
        packed = _all_reduce(torch.cat(moments, dim=0), sync_group)
        outs, saved = [], []
        off = 0
        for i in range(n):
            rm, rv, nbt, relu, momentum, eps = meta[i]
            x, w, b, r = xs[i], tensors[4 * i + 1], tensors[4 * i + 2], tensors[4 * i + 3]
            C = x.shape[1]
            mi = K.bn_finalize(packed[off:off + C + 1], SYNC_COUNT, eps, momentum, rm, rv, nbt)     # a row slice: contiguous
            off += C + 1
            r = None if r is None else r.contiguous()
            amax = K.amax_request(x)
            y = K.amax_attach(K.bn_apply(x, mi, w, b, r, relu, amax=amax), amax)
            outs.append(y)
            saved += [x, mi, w, b, y if (relu and r is not None) else None]
        ctx.meta = [(m[3], tensors[4 * i + 3] is not None) for i, m in enumerate(meta)]
        ctx.sync_group = sync_group
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        n = len(ctx.meta)
        red = []
        for i in range(n):
            x, mi, w, b, out = saved[5 * i:5 * i + 5]
            relu, has_res = ctx.meta[i]
            mode = 0 if not relu else (2 if has_res else 1)
            dy = dys[i].contiguous()
            sums, d_w, d_b, g = K.bn_bwd_reduce(dy, x, out, mi, w, b, mode)
            red.append((dy, mode, sums, d_w, d_b, g))
        packed = _all_reduce(torch.cat([r[2] for r in red], dim=0), ctx.sync_group)
        grads = [None, None]
        off = 0
        for i in range(n):
            x, mi, w, b, out = saved[5 * i:5 * i + 5]
            dy, mode, _, d_w, d_b, g = red[i]
            C = x.shape[1]
            dx = None
            if ctx.needs_input_grad[2 + 4 * i]:
                amax = K.amax_request(x)
                dx = K.amax_attach(K.bn_bwd_apply(g if mode == 2 else dy, x, mi, w, b, packed[off:off + C + 1],
                                                  SYNC_COUNT, mode == 1, amax=amax), amax)
            off += C + 1
            d_res = (g if mode == 2 else dy) if (ctx.meta[i][1] and ctx.needs_input_grad[2 + 4 * i + 3]) else None
            grads += [dx, d_w if w is not None else None, d_b if b is not None else None, d_res]
        return tuple(grads)


def bn_act_group(sites):
    """sites: list of (bn module, x, residual or None, relu or None). Returns the list of outputs. Sites whose module is
    not in synchronised training mode (single rank, eval) are evaluated one by one through the module itself."""
    groups = [bn._sync_group() if (bn.training and isinstance(bn, FusedSyncBatchNorm)) else None for bn, _, _, _ in sites]
    if len(sites) < 2 or any(g is None for g in groups) or any(g is not groups[0] for g in groups):
        return [bn(x, residual=r, relu=relu) for bn, x, r, relu in sites]
    meta, tensors = [], []
    for bn, x, r, relu in sites:
        if x.dim() != 4 or bn.momentum is None or not bn.track_running_stats:
            return [b_(x_, residual=r_, relu=l_) for b_, x_, r_, l_ in sites]
        relu = (bn.act == 'relu') if relu is None else bool(relu)
        meta.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, relu, float(bn.momentum), float(bn.eps)))
        tensors += [x, bn.weight, bn.bias, r]
    return list(_BNActGroup.apply(meta, groups[0], *tensors))


class _FusedMixin(object):
    """forward(x, residual=None, relu=None): relu=None -> the module's own default (`self.act == 'relu'`)."""
    act = None

    def _sync_group(self):
        return None

    def forward(self, x, residual=None, relu=None):
        if x.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
        relu = (self.act == 'relu') if relu is None else bool(relu)
        if self.momentum is None:
            raise NotImplementedError('cumulative moving average (momentum=None) is not implemented on the fused path')
        training = self.training or not self.track_running_stats
        track = self.track_running_stats
        return _BNAct.apply(x, self.weight, self.bias, residual,
                            self.running_mean if track else None, self.running_var if track else None,
                            self.num_batches_tracked if (track and self.training) else None,
                            training, relu, float(self.momentum), float(self.eps),
                            self._sync_group() if training else None)

    def extra_repr(self):
        return super(_FusedMixin, self).extra_repr() + ', act={}'.format(self.act)


class FusedBatchNorm2d(_FusedMixin, nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, act=None, **kw):
        nn.BatchNorm2d.__init__(self, num_features, eps, momentum, affine, track_running_stats, **kw)
        self.act = act


class FusedSyncBatchNorm(_FusedMixin, nn.SyncBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None,
                 act=None, **kw):
        nn.SyncBatchNorm.__init__(self, num_features, eps, momentum, affine, track_running_stats, process_group, **kw)
        self.act = act

    def _sync_group(self):
        if not D.is_distributed():
            return None                      # no process group: plain batch statistics, like nn.SyncBatchNorm
        import torch.distributed as dist
        group = self.process_group if self.process_group is not None else dist.group.WORLD
        return group if (dist.get_world_size(group) > 1 or D.exercise_single_rank()) else None
