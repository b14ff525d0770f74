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


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, num_batches_tracked, training, relu,
                momentum, eps, sync_group):
        """sync_group: None = local statistics; otherwise the process group whose ranks share statistics."""
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        n_local = x.numel() // x.shape[1]
        count = float(n_local)
        if training:
            if sync_group is not None:
                world = torch.distributed.get_world_size(sync_group)
                moments = _all_reduce(K.bn_stats(x), sync_group)
                count = float(n_local * world)          # equal per-rank batch (data_loader.py:137 splits evenly)
                mi = K.bn_finalize(moments, count, eps, momentum, running_mean, running_var, num_batches_tracked)
                y = K.bn_apply(x, mi, weight, bias, residual, relu)
            else:
                # single rank: statistics + (finalise, running statistics, apply) in two launches
                y, mi = K.bn_fwd(x, weight, bias, residual, relu, eps, momentum, running_mean, running_var,
                                 num_batches_tracked)
        else:
            mi = torch.stack([running_mean, torch.rsqrt(running_var + eps)], dim=1).contiguous()
            y = K.bn_apply(x, mi, weight, bias, residual, relu)
        ctx.meta = (training, relu, residual is not None, count, sync_group)
        ctx.save_for_backward(x, mi, weight, bias, y if (relu and residual is not None) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mi, weight, bias, out = ctx.saved_tensors
        training, relu, has_res, count, sync_group = ctx.meta
        dy = dy.contiguous()
        mode = 0 if not relu else (2 if has_res else 1)
        if sync_group is None or not training:
            # single rank (or frozen statistics): reduce + (sums, parameter gradients, dx) in two launches
            dx, d_weight, d_bias, g = K.bn_bwd(dy, x, out, mi, weight, bias, mode, training, ctx.needs_input_grad[0])
        else:
            sums, d_weight, d_bias, g = K.bn_bwd_reduce(dy, x, out, mi, weight, bias, mode)
            dx = None
            if ctx.needs_input_grad[0]:
                sums = _all_reduce(sums, sync_group)
                dx = K.bn_bwd_apply(g if mode == 2 else dy, x, mi, weight, bias, sums, count, mode == 1)
        d_res = None
        if has_res:
            d_res = g if mode == 2 else dy          # the add passes the (masked) gradient straight through
        return (dx, d_weight if weight is not None else None, d_bias if bias is not None else None,
                d_res if ctx.needs_input_grad[3] else None, None, None, None, None, None, None, None, None)


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
        return group if dist.get_world_size(group) > 1 else None
