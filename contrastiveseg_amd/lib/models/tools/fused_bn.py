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
        if training and hasattr(K, 'mark_zero_channel_sum'):
            K.mark_zero_channel_sum(dx)          # batch statistics: sum(dx) = 0 per channel -> the bias gradient of the convolution in front
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


# ----------------------------------------------------------------------------------------------------------
# Round 5: the residual blocks of ONE depth of HRNet's parallel branches under SyncBN as ONE autograd node.
# Reference shape of the work: BasicBlock.forward (lib/models/backbones/hrnet/hrnet_backbone.py:49-65) of n = 2..4 branches that have
# no data dependence on each other; nn.SyncBatchNorm would issue 2 n collectives per direction here, _BNActGroup issues 2, and the
# lockstep form of HighResolutionModule used to build them from 2 n convolution nodes + 2 grouped-BN nodes. The multi-rank path is
# HOST-bound below ~4 images per GPU (profiles/r05_dist_single_rank.txt), so the nodes are what costs: this Function issues exactly the
# same library calls and collectives in the same order (bit-identical results, tests/test_distributed_gloo.py) behind one apply().
# ----------------------------------------------------------------------------------------------------------
def _group_exchange_forward(xs, bns, residuals, relu, sync_group):
    """Grouped SyncBN(+residual)+ReLU over independent sites (the body of _BNActGroup.forward) -> (outs, mean_invstd per site, max|y|
    records). ONE all-reduce of the concatenated [C_i + 1, 2] fp64 moments."""
    moments = []
    for x in xs:
        tiles = K.known_tile_stats(x)
        moments.append(K.bn_tiles_moments(tiles) if tiles is not None else K.bn_stats(x))
    packed = _all_reduce(torch.cat(moments, dim=0), sync_group)
    outs, mis, ams = [], [], []
    off = 0
    for x, bn, r in zip(xs, bns, residuals):
        C = x.shape[1]
        mi = K.bn_finalize(packed[off:off + C + 1], SYNC_COUNT, float(bn.eps), float(bn.momentum), bn.running_mean, bn.running_var,
                           bn.num_batches_tracked)
        off += C + 1
        amax = K.amax_request(x)
        outs.append(K.bn_apply(x, mi, bn.weight, bn.bias, r, relu, amax=amax))
        mis.append(mi)
        ams.append(amax)
    return outs, mis, ams


def _group_exchange_backward(dys, xs, outs, mis, bns, mode, sync_group):
    """Adjoint of _group_exchange_forward -> per site (dx, d_weight, d_bias, masked gradient or None, max|dx| record). ONE all-reduce of
    the concatenated gradient sums. mode: 1 = ReLU mask from x, 2 = from `out` (residual sites; the masked gradient is also the
    residual's)."""
    red = []
    for dy, x, out, mi, bn in zip(dys, xs, outs, mis, bns):
        red.append(K.bn_bwd_reduce(dy, x, out if mode == 2 else None, mi, bn.weight, bn.bias, mode))
    packed = _all_reduce(torch.cat([r[0] for r in red], dim=0), sync_group)
    res = []
    off = 0
    for dy, x, mi, bn, (_, d_w, d_b, g) in zip(dys, xs, mis, bns, red):
        C = x.shape[1]
        amax = K.amax_request(x)
        dx = K.bn_bwd_apply(g if mode == 2 else dy, x, mi, bn.weight, bn.bias, packed[off:off + C + 1], SYNC_COUNT, mode == 1, amax=amax)
        off += C + 1
        res.append((dx, d_w, d_b, g, amax))
    return res


def _grouped_sync_ok():
    """The grouped launches of round 6 (kernels.conv3x3_group_run / bn_group_sync_* / conv3x3_group_wrw) serve the SyncBN node too:
    per depth 8 launches + 2 collectives forward and 14 + 2 backward whatever the number of branches (before: 8 n and ~10 n launches).
    CSEG_BLOCK_GROUP=0 restores the per-member calls."""
    return getattr(K, "BLOCK_GROUP", False) and hasattr(K, "bn_group_sync_moments")


class BasicBlockGroupSync(torch.autograd.Function):
    """n residual blocks (conv3x3 -> SyncBN -> ReLU -> conv3x3 -> SyncBN -> + x -> ReLU) of one depth, statistics exchanged together.
    tensors: per block x, w1, g1, b1, w2, g2, b2."""

    @staticmethod
    def forward(ctx, blocks, sync_group, *tensors):
        n = len(blocks)
        xs = [tensors[7 * i].contiguous() for i in range(n)]
        if _grouped_sync_ok() and all(blk.conv1.weight.shape[0] >= 32 for blk in blocks):
            axs = [K.amax_of(x) for x in xs]
            c1s, _ = K.conv3x3_group_run([(x, blk.conv1.weight, False, ax, None) for x, blk, ax in zip(xs, blocks, axs)], want_stats=True)
            a1s, mi1, am1 = K.bn_group_sync_apply(c1s, [b.bn1 for b in blocks], [None] * n, True,
                                                  _all_reduce(K.bn_group_sync_moments(c1s), sync_group))
            c2s, _ = K.conv3x3_group_run([(a1, blk.conv2.weight, False, am, None) for a1, blk, am in zip(a1s, blocks, am1)], want_stats=True)
            outs, mi2, am2 = K.bn_group_sync_apply(c2s, [b.bn2 for b in blocks], xs, True,
                                                   _all_reduce(K.bn_group_sync_moments(c2s), sync_group))
            for o, am in zip(outs, am2):
                K.amax_attach(o, am)
            ctx.blocks, ctx.sync_group, ctx.nts, ctx.axs, ctx.am1 = blocks, sync_group, None, axs, am1
            ctx.save_for_backward(*(xs + c1s + a1s + c2s + outs + mi1 + mi2))
            return tuple(outs)
        nts, axs, c1s = [], [], []
        for blk, x in zip(blocks, xs):
            c = blk.conv1.weight.shape[0]
            nt = K.conv3x3_sb_pick_nt(x, c) if c in K.CONV3X3_SB_PICK_NT_CHANNELS else 0
            ax = K.amax_of(x)
            nts.append(nt)
            axs.append(ax)
            c1s.append(K.conv3x3_sb_run(x, blk.conv1.weight, False, None, nt, ax=ax, want_stats=True))
        a1s, mi1, am1 = _group_exchange_forward(c1s, [b.bn1 for b in blocks], [None] * n, True, sync_group)
        c2s = [K.conv3x3_sb_run(a1, blk.conv2.weight, False, None, nt, ax=am, want_stats=True)
               for a1, blk, nt, am in zip(a1s, blocks, nts, am1)]
        outs, mi2, am2 = _group_exchange_forward(c2s, [b.bn2 for b in blocks], xs, True, sync_group)
        for o, am in zip(outs, am2):
            K.amax_attach(o, am)
        ctx.blocks, ctx.sync_group, ctx.nts, ctx.axs, ctx.am1 = blocks, sync_group, nts, axs, am1
        ctx.save_for_backward(*(xs + c1s + a1s + c2s + outs + mi1 + mi2))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        blocks, sync_group, nts, axs, am1 = ctx.blocks, ctx.sync_group, ctx.nts, ctx.axs, ctx.am1
        n = len(blocks)
        sv = ctx.saved_tensors
        xs, c1s, a1s, c2s, outs, mi1, mi2 = (list(sv[k * n:(k + 1) * n]) for k in range(7))
        dys = [d.contiguous() for d in dys]
        if nts is None:                             # the forward ran on the grouped launches
            need = ctx.needs_input_grad
            packed, st = K.bn_group_sync_bwd_reduce(dys, c2s, outs, mi2, [b.bn2 for b in blocks], 2)
            r2 = K.bn_group_sync_bwd_apply(st, _all_reduce(packed, sync_group))
            da1s, _ = K.conv3x3_group_run([(r2[i][0], blocks[i].conv2.weight, True, r2[i][4], None) for i in range(n)])
            dw2s = K.conv3x3_group_wrw([(a1s[i], r2[i][0], am1[i], r2[i][4]) for i in range(n)])
            packed, st = K.bn_group_sync_bwd_reduce(da1s, c1s, [None] * n, mi1, [b.bn1 for b in blocks], 1)
            r1 = K.bn_group_sync_bwd_apply(st, _all_reduce(packed, sync_group))
            dxs, _ = K.conv3x3_group_run([(r1[i][0], blocks[i].conv1.weight, True, r1[i][4], r2[i][3]) for i in range(n)])
            dw1s = K.conv3x3_group_wrw([(xs[i], r1[i][0], axs[i], r1[i][4]) for i in range(n)])
            grads = [None, None]
            for i, blk in enumerate(blocks):
                grads += [dxs[i] if need[2 + 7 * i] else None, dw1s[i] if need[2 + 7 * i + 1] else None,
                          r1[i][1] if blk.bn1.weight is not None else None, r1[i][2] if blk.bn1.bias is not None else None,
                          dw2s[i] if need[2 + 7 * i + 4] else None,
                          r2[i][1] if blk.bn2.weight is not None else None, r2[i][2] if blk.bn2.bias is not None else None]
            return tuple(grads)
        r2 = _group_exchange_backward(dys, c2s, outs, mi2, [b.bn2 for b in blocks], 2, sync_group)
        da1s, dw2s = [], []
        for i, blk in enumerate(blocks):
            dc2, _, _, _, am = r2[i]
            da1s.append(K.conv3x3_sb_run(dc2, blk.conv2.weight, True, None, nts[i], ax=am))
            dw2s.append(K.conv3x3_sb_wrw(a1s[i], dc2, ax=am1[i], ady=am) if ctx.needs_input_grad[2 + 7 * i + 4] else None)
        r1 = _group_exchange_backward(da1s, c1s, [None] * n, mi1, [b.bn1 for b in blocks], 1, sync_group)
        grads = [None, None]
        for i, blk in enumerate(blocks):
            dc1, dg1, db1, _, amb = r1[i]
            _, dg2, db2, g, _ = r2[i]
            dx = K.conv3x3_sb_run(dc1, blk.conv1.weight, True, None, nts[i], ax=amb, addend=g) if ctx.needs_input_grad[2 + 7 * i] else None
            dw1 = K.conv3x3_sb_wrw(xs[i], dc1, ax=axs[i], ady=amb) if ctx.needs_input_grad[2 + 7 * i + 1] else None
            grads += [dx, dw1, dg1 if blk.bn1.weight is not None else None, db1 if blk.bn1.bias is not None else None, dw2s[i],
                      dg2 if blk.bn2.weight is not None else None, db2 if blk.bn2.bias is not None else None]
        return tuple(grads)


def basic_block_group(blocks, xs):
    """The blocks of one depth of parallel branches on inputs xs -> outputs, as ONE node where every block qualifies (split kernels in
    all three directions, SyncBN in training mode with one shared group, no downsample), else None (the caller takes the per-op path)."""
    if len(blocks) < 2 or not getattr(K, "BLOCK_FUSED", False) or not hasattr(K, "basic_block_split_ok"):
        return None
    group = None
    for blk, x in zip(blocks, xs):
        if blk.downsample is not None or blk.stride != 1 or not blk.training:
            return None
        for bn in (blk.bn1, blk.bn2):
            if not (isinstance(bn, FusedSyncBatchNorm) and bn.training and bn.track_running_stats and bn.momentum is not None
                    and bn.weight is not None):
                return None
            g = bn._sync_group()
            if g is None or (group is not None and g is not group):
                return None
            group = g
        if not (blk.conv1.bn_follows and blk.conv2.bn_follows and K.CONV_EPILOGUE_STATS and K.split_arith_id()
                and K.basic_block_split_ok(x, blk.conv1.weight, blk.conv2.weight)):
            return None
    tensors = []
    for blk, x in zip(blocks, xs):
        tensors += [x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
    return list(BasicBlockGroupSync.apply(tuple(blocks), group, *tensors))


class _BNActGroupLocal(torch.autograd.Function):
    """Round 6, single rank: independent BN(+ReLU) sites of one depth (the conv + BN paths of an HRNet exchange unit: reference
    lib/models/backbones/hrnet/hrnet_backbone.py:230-250 builds them, :271-286 loops over them) on the grouped launches -- statistics
    finalisation + apply forward, reduction + apply backward: two launches per direction for ALL sites instead of two per site
    (16 sites in a four-branch unit). Same kernels' bodies per site as _BNAct: bit-identical. tensors: per site x, weight, bias."""

    @staticmethod
    def forward(ctx, bns, relu, *tensors):
        n = len(bns)
        xs = [tensors[3 * i].contiguous() for i in range(n)]
        ys, mis, ams = K.bn_group_fwd(xs, bns, [None] * n, relu)
        for y, am in zip(ys, ams):
            K.amax_attach(y, am)
        ctx.bns, ctx.relu = bns, relu
        ctx.save_for_backward(*(xs + mis))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        bns, n = ctx.bns, len(ctx.bns)
        sv = ctx.saved_tensors
        xs, mis = list(sv[:n]), list(sv[n:])
        res = K.bn_group_bwd([d.contiguous() for d in dys], xs, [None] * n, mis, bns, 1 if ctx.relu else 0)
        grads = [None, None]
        for (dx, dw, db, _, am), bn in zip(res, bns):
            K.amax_attach(dx, am)
            grads += [dx, dw if bn.weight is not None else None, db if bn.bias is not None else None]
        return tuple(grads)


def _bn_act_group_local(sites):
    """Single-rank training-mode sites whose inputs carry the producing convolution's epilogue statistics, grouped by their ReLU flag;
    -> outputs in the order of `sites`, or None when the grouped launches do not apply (the caller evaluates the sites one by one)."""
    if not (getattr(K, "BLOCK_GROUP", False) and hasattr(K, "bn_group_fwd") and hasattr(K, "known_tile_stats")):
        return None
    flags = []
    for bn, x, r, relu in sites:
        if not (isinstance(bn, _FusedMixin) and bn.training and bn.track_running_stats and bn.momentum is not None and r is None
                and x.dim() == 4 and bn._sync_group() is None and K.known_tile_stats(x) is not None and x.requires_grad):
            return None
        flags.append((bn.act == 'relu') if relu is None else bool(relu))
    outs = [None] * len(sites)
    for flag in (True, False):
        idx = [i for i, f in enumerate(flags) if f == flag]
        for lo in range(0, len(idx), 8):                  # CSEG_GROUP_MAX members per launch
            part = idx[lo:lo + 8]
            if len(part) == 1:
                bn, x, r, relu = sites[part[0]]
                outs[part[0]] = bn(x, residual=r, relu=relu)
            elif part:
                tensors = []
                for i in part:
                    tensors += [sites[i][1], sites[i][0].weight, sites[i][0].bias]
                for i, y in zip(part, _BNActGroupLocal.apply(tuple(sites[i][0] for i in part), flag, *tensors)):
                    outs[i] = y
    return outs


def bn_act_group(sites):
    """sites: list of (bn module, x, residual or None, relu or None). Returns the list of outputs. Sites whose module is
    not in synchronised training mode (single rank, eval) are evaluated one by one through the module itself -- or, single rank in
    training mode with epilogue statistics (round 6), together on the grouped launches."""
    groups = [bn._sync_group() if (bn.training and isinstance(bn, FusedSyncBatchNorm)) else None for bn, _, _, _ in sites]
    if len(sites) >= 2 and all(g is None for g in groups):
        local = _bn_act_group_local(sites)
        if local is not None:
            return local
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
