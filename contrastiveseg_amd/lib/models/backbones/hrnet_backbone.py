"""HRNetV2 encoder (W18/W32/W48/W64) with the reference's parameter names and construction order
(lib/models/backbones/hrnet/hrnet_backbone.py:35-574, configs hrnet_config.py:46-73), so reference checkpoints
load unchanged and the same torch seed gives identical random initialisation.

The network is expressed as data (stage table below) plus three small module kinds; convolutions execute on
MIOpen through PyTorch-ROCm, every BatchNorm (+ the ReLU / residual add behind it) on the fused cseg_bn_* kernels
(lib/models/tools/fused_bn.py); the cross-resolution exchange (sum of same-resolution terms + bilinear-upsampled
coarse terms + ReLU, reference :271-286) is one HIP kernel per output branch (cseg_fuse_sum_fwd/bwd).
Factory keys follow lib/models/backbones/hrnet/hrnet_backbone.py:742-803 ('hrnet18' ... 'hrnet64'); BN is hard-wired
to torch SyncBN with momentum 0.1 exactly as the reference factory does (:773)."""
import torch.nn as nn

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.models.tools.fused_bn import basic_block_group, bn_act_group
from contrastiveseg_amd.lib.models.tools.module_helper import Conv1x1, Conv3x3, ModuleHelper, StemConv3x3

# width -> per-stage (modules, blocks per branch); channel list is width * (1, 2, 4, 8)[:branches]
STAGES = {2: (1, 4), 3: (4, 4), 4: (3, 4)}
WIDTHS = {'hrnet18': 18, 'hrnet32': 32, 'hrnet48': 48, 'hrnet64': 64}


def _norm(bn_type, c, momentum, act=None):
    return ModuleHelper.BatchNorm2d(bn_type=bn_type)(c, momentum=momentum, act=act)


def _conv_bn(cin, cout, k, stride, bn_type, momentum, relu):
    """conv -> BN [-> ReLU]; the ReLU (stateless third child in the reference) is fused into the norm kernel."""
    if k == 1 and stride == 1:
        conv = Conv1x1(cin, cout, bias=False)
    elif k == 3 and stride in (1, 2):
        conv = Conv3x3(cin, cout, stride)                # an nn.Conv2d(cin, cout, 3, stride, 1, bias=False); split kernels where covered
    else:
        conv = nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False)
    return nn.Sequential(conv,
                         _norm(bn_type, cout, momentum, 'relu' if relu else None))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_type=None, bn_momentum=0.1):
        super(BasicBlock, self).__init__()
        self.conv1 = Conv3x3(inplanes, planes, stride)      # an nn.Conv2d; MFMA kernel on the narrow branches
        self.bn1 = _norm(bn_type, planes, bn_momentum)
        self.conv2 = Conv3x3(planes, planes)
        self.bn2 = _norm(bn_type, planes, bn_momentum)
        self.downsample = downsample
        self.stride = stride

    def _fused_route(self, x):
        """The whole block as one autograd node (kernels.BasicBlockSplit) where every convolution direction runs on the split kernels
        and the BatchNorms take plain single-rank batch statistics; decided once per input shape and switch setting."""
        fn = getattr(K, "basic_block_split_ok", None)
        if fn is None or self.downsample is not None or not self.training:
            return False
        # everything the decision reads is in the key: a BatchNorm frozen on its own (`bn.eval()` inside a training block),
        # a process group / CSEG_DIST_SINGLE_RANK that appears after the first forward, a stride -- each changes the route at once
        bn1, bn2 = self.bn1, self.bn2
        g1, g2 = getattr(bn1, "_sync_group", None), getattr(bn2, "_sync_group", None)
        g1, g2 = (g1() if g1 is not None else None), (g2() if g2 is not None else None)
        key = (x.shape, x.requires_grad, x.is_contiguous(), K.BLOCK_FUSED, K.CONV3X3_SPLIT_BF16, K.CONV3X3_SB_WRW, K.CONV3X3_FORK, K.SPLIT_ARITH,
               K.CONV3X3_SB_MIN_TILES, bn1.training, bn2.training, id(g1), id(g2), self.stride)
        hit = self.__dict__.get("_route")
        if hit is None or hit[0] != key:
            bn_ok = all(b.training and b.track_running_stats and b.momentum is not None and g is None and b.weight is not None
                        for b, g in ((bn1, g1), (bn2, g2))) and self.stride == 1
            hit = (key, bool(bn_ok and fn(x, self.conv1.weight, self.conv2.weight)))
            self.__dict__["_route"] = hit
        return hit[1]

    def forward(self, x):
        if self._fused_route(x):
            return K.basic_block_split(x, self)
        if self.downsample is None:
            out, res = self.conv1.forward_fork(x)                     # conv1(x) and the identity path from one autograd node
        else:
            out, res = self.conv1(x), self.downsample(x)
        out = self.bn1(out, relu=True)                                # BN + ReLU: one statistics pass + one apply pass
        return self.bn2(self.conv2(out), residual=res, relu=True)     # BN + residual add + ReLU in the same apply pass


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_type=None, bn_momentum=0.1):
        super(Bottleneck, self).__init__()
        self.conv1 = Conv1x1(inplanes, planes, bias=False)
        self.bn1 = _norm(bn_type, planes, bn_momentum)
        self.conv2 = Conv3x3(planes, planes, stride)        # an nn.Conv2d (same init, same state_dict); 64 ch -> split kernel
        self.bn2 = _norm(bn_type, planes, bn_momentum)
        self.conv3 = Conv1x1(planes, planes * 4, bias=False)
        self.bn3 = _norm(bn_type, planes * 4, bn_momentum)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.downsample is None:
            # conv1 and the skip connection as one autograd node: their two gradients meet in the epilogue of conv1's backward-data kernel
            # (module_helper.Conv1x1.forward_skip) instead of autograd's add over two 268 MB tensors per block
            c1, res = self.conv1.forward_skip(x)
        else:
            c1, res = self.conv1(x), self.downsample(x)
        out = self.bn1(c1, relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=res, relu=True)


def _block_chain(block, inplanes, planes, n, bn_type, momentum):
    """downsample projection (created first, as the reference does) + n residual blocks"""
    down = None
    if inplanes != planes * block.expansion:
        down = _conv_bn(inplanes, planes * block.expansion, 1, 1, bn_type, momentum, relu=False)
    blocks = [block(inplanes, planes, 1, down, bn_type=bn_type, bn_momentum=momentum)]
    for _ in range(1, n):
        blocks.append(block(planes * block.expansion, planes, bn_type=bn_type, bn_momentum=momentum))
    return nn.Sequential(*blocks)


_FORK_STREAMS = {}


FORK_PRIORITY = None        # set from CSEG_FORK_PRIORITY on first use: HIP stream priority of the side streams (0 = default)


def _fork_streams(device, n):
    import torch
    global FORK_PRIORITY
    if FORK_PRIORITY is None:
        FORK_PRIORITY = int(_os.environ.get("CSEG_FORK_PRIORITY", "0"))
    have = _FORK_STREAMS.setdefault((device.type, device.index), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device, priority=FORK_PRIORITY))
    return have[:n]


import os as _os

# Forked branches / exchange paths in EAGER steps too (default on; CSEG_BRANCH_STREAMS=0 = one stream). One host thread can feed the
# four queues because the kernels of a batch-8 step are long: measured on the MI355X (profiles/r04_branch_streams_ab.txt) 96.7-100.7 ms
# per step on one stream against 89.7-95.7 with the branches forked, same losses. Only on the GPU (there are no streams elsewhere).
EAGER_FORKS = _os.environ.get("CSEG_BRANCH_STREAMS", "1") == "1"


def _capturing():
    from contrastiveseg_amd.segmentor.tools import step_graph
    return step_graph.capturing()


# Eager forks only where the kernels are long enough for one host thread to keep four queues busy: the finest branch at least
# 4 x 128 x 256 pixels (measured, profiles/r04_bench_b2_eager.json vs profiles/r04_step_graph_ab.txt: at 2 images per GPU the forks cost
# 65.4 vs 57.1 ms per step -- their wait / record calls land on a host that is the limit there anyway).
EAGER_FORK_MIN_PIXELS = int(_os.environ.get("CSEG_BRANCH_STREAMS_MIN_PIXELS", str(4 * 128 * 256)))


def _capture_forks(x=None):
    from contrastiveseg_amd.segmentor.tools import step_graph
    if x is not None and not x.is_cuda:
        return False
    if step_graph.capturing():
        return step_graph.BRANCH_STREAMS
    if not EAGER_FORKS or (x is not None and x.shape[0] * x.shape[2] * x.shape[3] < EAGER_FORK_MIN_PIXELS):
        return False
    # Under a process group: DDP's reducer launches a bucket's all-reduce relative to the stream of the hook that COMPLETES the bucket
    # and would not wait for gradients still being written on the other fork streams -- unless the wrapper carries the comm hook of
    # segmentor/tools/module_runner.py (join_fork_streams before every bucket's collective), which sets DDP_FORKS_OK. (SyncBN models
    # never get here: their branches run in lockstep around the batched statistics exchange, HighResolutionModule.forward.)
    from contrastiveseg_amd.lib.utils.distributed import is_distributed
    return DDP_FORKS_OK or not is_distributed()


EXCHANGE_GROUPED = _os.environ.get("CSEG_EXCHANGE_GROUPED", "1") == "1"      # round 6: see HighResolutionModule.forward
LOCKSTEP_GROUP_NODE = _os.environ.get("CSEG_LOCKSTEP_GROUP_NODE", "1") == "1"      # fused_bn.BasicBlockGroupSync (round 5)
DDP_FORKS_OK = False        # set by ModuleRunner._make_parallel once the DDP wrapper joins the fork streams before its collectives


def join_fork_streams(device):
    """The current stream waits for everything enqueued so far on the fork streams of `device` (a no-op when no fork was ever taken).
    What a consumer that does not know about the forks needs before it reads gradients: DDP's bucket all-reduce (comm hook), a
    gradient clip right after backward()."""
    import torch
    have = _FORK_STREAMS.get((device.type, device.index))
    if have:
        cur = torch.cuda.current_stream(device)
        for s_ in have:
            cur.wait_stream(s_)


class HighResolutionModule(nn.Module):
    """One multi-resolution exchange unit: per-branch residual chains, then every output resolution sums all
    branches (strided 3x3 chains going down, 1x1 + bilinear going up). Reference :108-288."""

    def __init__(self, channels, num_blocks, bn_type, bn_momentum, multi_scale_output=True):
        super(HighResolutionModule, self).__init__()
        nb = len(channels)
        self.num_branches = nb
        self.branches = nn.ModuleList([_block_chain(BasicBlock, c, c, num_blocks, bn_type, bn_momentum)
                                       for c in channels])
        fuse = []
        for i in range(nb if multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(_conv_bn(channels[j], channels[i], 1, 1, bn_type, bn_momentum, relu=False))
                elif j == i:
                    row.append(None)
                else:
                    steps = [_conv_bn(channels[j], channels[j], 3, 2, bn_type, bn_momentum, relu=True)
                             for _ in range(i - j - 1)]
                    steps.append(_conv_bn(channels[j], channels[i], 3, 2, bn_type, bn_momentum, relu=False))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse) if nb > 1 else None

    def _sync_active(self):
        bn = self.branches[0][0].bn1
        return self.training and getattr(bn, '_sync_group', lambda: None)() is not None

    def _branches_lockstep(self, x):
        """The branches are independent residual chains of equal length: run them block by block side by side, so that
        the BN sites of one depth share ONE statistics all-reduce per direction (fused_bn.bn_act_group)."""
        x = list(x)
        # (Round 5 also had an opt-in that forked the convolutions of a depth onto side streams, CSEG_LOCKSTEP_FORKS: measured slower inside
        # a one-rank RCCL group -- 118.3 vs 111.9 ms at batch 8, profiles/r05_dist_single_rank.txt -- and superseded by the grouped launches
        # of round 6, which run a depth's convolutions as ONE kernel; removed.)
        for k in range(len(self.branches[0])):
            blocks = [branch[k] for branch in self.branches]
            if LOCKSTEP_GROUP_NODE:
                grouped = basic_block_group(blocks, x)        # the whole depth as ONE autograd node where every block qualifies
                if grouped is not None:
                    x = grouped
                    continue
            c1 = [blk.conv1(xi) for blk, xi in zip(blocks, x)]
            y1 = bn_act_group([(blk.bn1, c, None, True) for blk, c in zip(blocks, c1)])
            c2 = [blk.conv2(y) for blk, y in zip(blocks, y1)]
            res = [xi if blk.downsample is None else blk.downsample(xi) for blk, xi in zip(blocks, x)]
            x = bn_act_group([(blk.bn2, c, r, True) for blk, c, r in zip(blocks, c2, res)])
        return x

    def _exchange_lockstep(self, x):
        """All (output i <- input j) conv+BN paths of the exchange unit, advanced one conv+BN at a time across paths."""
        chains = {}                                   # (i, j) -> list of Sequential(conv, bn)
        for i, row in enumerate(self.fuse_layers):
            for j in range(self.num_branches):
                if j == i:
                    continue
                chains[(i, j)] = [row[j]] if j > i else list(row[j])
        take = self._fan(x)
        cur = {key: take(key[1], key[0]) for key in chains}
        for depth in range(max(len(c) for c in chains.values())):
            keys = [key for key, c in chains.items() if len(c) > depth]
            convs = [chains[key][depth][0](cur[key]) for key in keys]
            outs = bn_act_group([(chains[key][depth][1], c, None, None) for key, c in zip(keys, convs)])
            for key, o in zip(keys, outs):
                cur[key] = o
        outs = []
        for i in range(len(self.fuse_layers)):
            same = [take(j, i) if j == i else cur[(i, j)] for j in range(i + 1)]
            low = [cur[(i, j)] for j in range(i + 1, self.num_branches)]
            outs.append(K.fuse_sum_relu(same, low))
        return outs

    def _fan(self, x):
        """-> take(j, i): branch j as output i reads it. With K.FANOUT_SUM every output gets its own alias behind one autograd node
        per branch (one gradient sum instead of up to three `add` launches, kernels.fan_out); otherwise the tensor itself."""
        n_out = len(self.fuse_layers)
        if K.FANOUT_SUM and n_out > 1:
            xf = [K.fan_out(xj, n_out) for xj in x]
            return lambda j, i: xf[j][i]
        return lambda j, i: x[j]

    def _branches_forked(self, x):
        """Inside a hipGraph capture (segmentor/tools/step_graph.py): every branch after the first on a side stream of its own, forked
        from and joined to the capturing stream -- the captured graph then has one independent path per branch (four residual chains of
        equal flops whose kernels fill 4 / 1 / 1 / 0.5 rounds of the 256 CUs when they run one after the other), and autograd replays the
        same fork in backward (a node's backward runs on the stream of its forward). Eager runs keep one stream: the host could not feed
        four anyway."""
        import torch
        cur = torch.cuda.current_stream(x[0].device)
        streams = _fork_streams(x[0].device, len(self.branches) - 1)
        outs = [None] * len(self.branches)
        for i in range(1, len(self.branches)):
            streams[i - 1].wait_stream(cur)
            x[i].record_stream(streams[i - 1])        # allocated on `cur`, read (now and again in backward) on the side stream
            with torch.cuda.stream(streams[i - 1]):
                outs[i] = self.branches[i](x[i])
        outs[0] = self.branches[0](x[0])
        for i, s in enumerate(streams):
            cur.wait_stream(s)
            outs[i + 1].record_stream(cur)            # allocated on the side stream, read by the exchange unit on `cur`
        return outs

    def _exchange_forked(self, x):
        """The exchange unit with one stream per OUTPUT resolution (see _branches_forked): output i's incoming paths -- 1x1 conv + BN
        from the coarser branches, chains of stride-2 convolutions from the finer ones -- and its fused sum run on stream i; the paths
        of different outputs share nothing but their inputs. These are the small launches of the step (<= 256 blocks each)."""
        import torch
        cur = torch.cuda.current_stream(x[0].device)
        streams = _fork_streams(x[0].device, len(self.fuse_layers) - 1)
        take = self._fan(x)

        def row_out(i):
            row = self.fuse_layers[i]
            same = [take(j, i) if j == i else row[j](take(j, i)) for j in range(i + 1)]
            low = [row[j](take(j, i)) for j in range(i + 1, self.num_branches)]
            return K.fuse_sum_relu(same, low)

        outs = [None] * len(self.fuse_layers)
        # Every side stream first waits for what is on the calling stream NOW (the branch outputs), then output 0 is enqueued on the
        # calling stream and the others on their streams -- in the order 0, 1, 2, ... of the single-stream path, so that the autograd
        # nodes are created in the same order and backward accumulates the (up to four) gradients that meet at a branch output in the
        # same order: a forked step is then BIT-identical to the single-stream one. (Round 6: with outputs 1.. created before output 0 the
        # fp32 sums differed in the last bit, and one SGD step of this network amplifies that to 1e-3 of the next loss -- the contrastive
        # term's anchor mining is an argmax, tests/test_gpu_streams.py.)
        for i in range(1, len(self.fuse_layers)):
            s = streams[i - 1]
            s.wait_stream(cur)
            for xj in x:
                xj.record_stream(s)                   # every output reads every branch
        outs[0] = row_out(0)
        for i in range(1, len(self.fuse_layers)):
            with torch.cuda.stream(streams[i - 1]):
                outs[i] = row_out(i)
        for i, s in enumerate(streams):
            cur.wait_stream(s)
            outs[i + 1].record_stream(cur)
        return outs

    def _branches_grouped(self, x):
        """Round 6, single rank: the branches advance depth by depth, every depth ONE autograd node on the grouped launches
        (kernels.BasicBlockGroup: the convolutions / BatchNorm passes / weight gradients of all n branches in one kernel each) -- the
        reference's loop over the branches (hrnet_backbone.py:262-288) turned inside out. Returns None when a block of the first depth
        does not qualify (eval mode, frozen statistics, strict-fp32 arithmetic, CSEG_BLOCK_GROUP=0 ...): the caller then takes the
        per-branch path."""
        group = getattr(K, "basic_block_group", None)
        if group is None or not getattr(K, "BLOCK_GROUP", False):
            return None
        x = list(x)
        for k in range(len(self.branches[0])):
            blocks = [branch[k] for branch in self.branches]
            out = group(blocks, x) if all(blk._fused_route(xi) for blk, xi in zip(blocks, x)) else None
            if out is None:
                if k == 0:
                    return None
                out = [blk(xi) for blk, xi in zip(blocks, x)]
            x = out
        return x

    def forward(self, x):
        sync = self._sync_active()
        grouped = None
        if not sync and self.num_branches > 1 and self.training:
            grouped = self._branches_grouped(x)
        if grouped is not None:
            x = grouped
        elif sync and self.num_branches > 1:
            x = self._branches_lockstep(x)
        elif self.num_branches > 1 and _capture_forks(x[0]):
            x = self._branches_forked(x)
        else:
            x = [branch(xi) for branch, xi in zip(self.branches, x)]
        if self.num_branches == 1:
            return x
        if sync:
            return self._exchange_lockstep(x)
        if grouped is not None and EXCHANGE_GROUPED and (_capturing() or not _capture_forks(x[0])):
            # Round 6: where the exchange unit runs on ONE stream anyway (below four images per GPU the forks do not pay, under a
            # hipGraph capture they crash), it advances depth by depth with the BatchNorm sites of a depth on the grouped launches
            # (fused_bn._BNActGroupLocal): two launches per depth and direction instead of two per site -- 230 dispatches per step less.
            # At batch 8 the forked form below stays: its four streams overlap the unit's small convolutions, which the depth-by-depth
            # form serialises (A/B/A/B on one MI355X: 87.9 / 87.2 ms grouped against 84.7 / 85.7 forked, profiles/r06_ab_exchange.txt).
            return self._exchange_lockstep(x)
        if _capture_forks(x[0]) and len(self.fuse_layers) > 1 and not _capturing():
            # (eager only: with these forks inside a hipGraph capture the END of the backward capture crashed on ROCm 7.2 -- GPU call
            # r04j12 -- while the branch forks alone capture fine; the replay keeps the exchange unit on the capturing stream)
            return self._exchange_forked(x)
        outs = []
        take = self._fan(x)
        for i, row in enumerate(self.fuse_layers):
            same = [take(j, i) if j == i else row[j](take(j, i)) for j in range(i + 1)]     # finer branches arrive strided
            low = [row[j](take(j, i)) for j in range(i + 1, self.num_branches)]             # coarser ones: 1x1 conv + BN
            outs.append(K.fuse_sum_relu(same, low))                                          # summed in branch order, then ReLU
        return outs


class HighResolutionNet(nn.Module):
    def __init__(self, width, bn_type='torchsyncbn', bn_momentum=0.1):
        super(HighResolutionNet, self).__init__()
        self.conv1 = StemConv3x3(64)                      # an nn.Conv2d(3, 64, 3, 2, 1, bias=False); fp32 stem kernels since round 6
        self.bn1 = _norm(bn_type, 64, bn_momentum)
        self.conv2 = Conv3x3(64, 64, 2)                  # an nn.Conv2d(64, 64, 3, 2, 1, bias=False) (hrnet_backbone.py:519 of the reference); split stride-2 kernels since round 6
        self.bn2 = _norm(bn_type, 64, bn_momentum)
        self.layer1 = _block_chain(Bottleneck, 64, 64, 4, bn_type, bn_momentum)
        prev = [256]
        for s in (2, 3, 4):
            chans = [width * (2 ** k) for k in range(s)]
            setattr(self, 'transition%d' % (s - 1), self._transition(prev, chans, bn_type, bn_momentum))
            n_mod, n_blk = STAGES[s]
            setattr(self, 'stage%d' % s, nn.Sequential(*[HighResolutionModule(chans, n_blk, bn_type, bn_momentum)
                                                         for _ in range(n_mod)]))
            prev = chans
        self.num_features = sum(prev)

    @staticmethod
    def _transition(prev, cur, bn_type, momentum):
        layers = []
        for i, c in enumerate(cur):
            if i < len(prev):
                layers.append(None if c == prev[i] else _conv_bn(prev[i], c, 3, 1, bn_type, momentum, relu=True))
            else:
                steps = []
                for j in range(i + 1 - len(prev)):
                    cout = c if j == i - len(prev) else prev[-1]
                    steps.append(_conv_bn(prev[-1], cout, 3, 2, bn_type, momentum, relu=True))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        x = self.bn2(self.conv2(x), relu=True)
        x = self.layer1(x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, 'transition%d' % (s - 1))
            xs = []
            for i, t in enumerate(trans):
                if t is None:
                    xs.append(ys[i])
                else:
                    xs.append(t(ys[-1]) if i >= len(ys) or s > 2 else t(ys[i]))
            ys = getattr(self, 'stage%d' % s)(xs)
        return ys


class HRNetBackbone(object):
    def __init__(self, configer):
        self.configer = configer

    def __call__(self):
        arch = self.configer.get('network', 'backbone')
        if arch not in WIDTHS:
            raise Exception('Architecture undefined!')
        net = HighResolutionNet(WIDTHS[arch], bn_type='torchsyncbn', bn_momentum=0.1)
        if self.configer.get('network', 'resume') is None:
            net = ModuleHelper.load_model(net, pretrained=self.configer.get('network', 'pretrained'),
                                          all_match=False, network='hrnet')
        return net
