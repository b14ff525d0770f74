"""ModuleRunner with the reference's responsibilities (segmentor/tools/module_runner.py:27-289): run counters in
the Configer, device placement, data-parallel wrap, resume, checkpoint save, lr warm-up.

Parallel wrap, MI355X-first: one process per GPU, torch DistributedDataParallel on the 'nccl' (= RCCL) backend
over xGMI. Gradient buckets are reduced on RCCL's own HIP stream while backward keeps producing gradients;
`gradient_as_bucket_view` removes the grad->bucket copy; the bucket size is a config knob
(network.ddp_bucket_mb, default 64 MB: xGMI rings are per-link bound, so fewer, larger messages than the
25 MB NVSwitch-era default). With the shipped model/criterion pairs all parameters receive gradients every step (the
`loss + 0 * loss_contrast` trick of the reference keeps the projection head in the graph during warm-up), so the
unused-parameter graph walk the reference always enables (:66-71) is switched on only where it is needed: when the
model emits an auxiliary output (`seg_aux`: DeepLab's DSN head, the OCR aux head) that the configured criterion does
not consume, or when `network.ddp_find_unused` says so. There is no single-process multi-GPU DataParallel path."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from contrastiveseg_amd.lib.utils.distributed import device_index, get_rank, is_distributed
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class ModuleRunner(object):
    def __init__(self, configer):
        self.configer = configer
        for key, val in (('iters', 0), ('last_iters', 0), ('epoch', 0), ('last_epoch', 0), ('max_performance', 0.0),
                         ('performance', 0.0), ('min_val_loss', 9999.0), ('val_loss', 9999.0)):
            if not self.configer.exists(key):
                self.configer.add([key], val)
        if not self.configer.exists('network', 'bn_type'):
            self.configer.add(['network', 'bn_type'], 'torchbn')
        Log.info('BN Type is {}.'.format(self.configer.get('network', 'bn_type')))

    def device(self):
        if torch.cuda.is_available() and not (self.configer.exists('gpu') and self.configer.get('gpu') is None
                                              and not is_distributed()):
            return torch.device('cuda', device_index() if is_distributed() else torch.cuda.current_device())
        return torch.device('cpu')

    def to_device(self, *params, force_list=False):
        dev = self.device()
        out = [p.to(dev) for p in params]
        return out if force_list or len(out) != 1 else out[0]

    def _make_parallel(self, net):
        if not is_distributed():
            return net
        bucket_mb = 64
        if self.configer.exists('network', 'ddp_bucket_mb'):
            bucket_mb = self.configer.get('network', 'ddp_bucket_mb')
        find_unused = self._has_unused_parameters()
        if self.configer.exists('network', 'ddp_find_unused'):
            find_unused = bool(self.configer.get('network', 'ddp_find_unused'))
        kwargs = dict(find_unused_parameters=find_unused, gradient_as_bucket_view=True, bucket_cap_mb=bucket_mb,
                      # no per-forward buffer broadcast (the reference's default, module_runner.py:62-76): SyncBN keeps the BN
                      # buffers identical on all ranks, and the memory queues are updated identically on every rank from the
                      # all-gathered keys (Trainer._enqueue_global) instead of being overwritten by rank 0's copy each step
                      # (2 x 97 MB per step at memory_size 5000). DDP's constructor still syncs the initial state from rank 0.
                      broadcast_buffers=False)
        if next(net.parameters()).is_cuda:
            kwargs.update(device_ids=[device_index()], output_device=device_index())
        ddp = torch.nn.parallel.DistributedDataParallel(net, **kwargs)
        # that argument only holds while every norm layer IS a SyncBN. With bn_type torchbn / FusedBatchNorm2d under DDP
        # each rank would drift to its own running statistics (and the rank-0 checkpoint would carry rank 0's only, unannounced).
        # The reference's behaviour for those layers -- rank 0's buffers win before every forward (DDP's default
        # broadcast_buffers=True, reference module_runner.py:62-71) -- is kept for exactly those buffers, a few hundred KB, without
        # bringing the 2 x 97 MB queue broadcast back.
        if next(net.parameters()).is_cuda:
            self._join_forks_before_collectives(ddp)
        unsynced = self.unsynced_norm_buffers(net)
        if unsynced:
            Log.info('DDP: {} running-statistics buffers of non-synchronised norm layers follow rank 0 before every forward.'
                     .format(len(unsynced)))
            # only in training mode -- the buffers cannot change during evaluation, and a collective per eval forward would
            # hang (or pair up with the metric's all-reduce) as soon as the ranks run different numbers of validation batches
            ddp.register_forward_pre_hook(lambda mod, _args: self.broadcast_from_rank0(unsynced) if mod.training else None)
        return ddp

    @staticmethod
    def _join_forks_before_collectives(ddp):
        """the forked HRNet branches / exchange paths (lib/models/backbones/hrnet_backbone.py) under DDP. The reducer
        issues a bucket's all-reduce from the autograd hook of the LAST gradient that lands in the bucket, ordered after the stream that
        hook runs on -- gradients of the same bucket written on the other fork streams were not waited for, which is why round 4 switched
        the forks off under any process group. The comm hook below makes the hook's stream wait for every fork stream first (all
        producers of the bucket have been ENQUEUED by then: a bucket completes only after each of its gradients was marked ready by its
        own hook), then runs DDP's own all-reduce (mean over ranks). One wait per fork stream per bucket: ~5 buckets x 3 streams per
        step. Reference: segmentor/tools/module_runner.py:62-76 (plain DDP on one stream)."""
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB

        def hook(state, bucket):
            HB.join_fork_streams(bucket.buffer().device)
            return default_hooks.allreduce_hook(state, bucket)

        ddp.register_comm_hook(None, hook)
        HB.DDP_FORKS_OK = True

    @staticmethod
    def unsynced_norm_buffers(net):
        """Buffers of norm layers that keep running statistics but do not synchronise them across ranks."""
        norm = torch.nn.modules.batchnorm._NormBase
        out = []
        for m in net.modules():
            if isinstance(m, norm) and m.track_running_stats and not isinstance(m, torch.nn.SyncBatchNorm):
                out += [b for b in m.buffers(recurse=False)]
        return out

    @staticmethod
    def broadcast_from_rank0(tensors, bucket_bytes=32 << 20):
        """What DDP's own buffer synchronisation does (torch/nn/parallel/distributed.py: _sync_buffers), for the given tensors."""
        if tensors and is_distributed():
            with torch.no_grad():
                torch.distributed._broadcast_coalesced(torch.distributed.group.WORLD, tensors, bucket_bytes, 0)

    def _has_unused_parameters(self):
        """True when the model has a head whose output the configured criterion never reads (its parameters would get
        no gradient and DDP's reducer would wait for them forever): models that emit `seg_aux` paired with a criterion
        without an auxiliary term. deeplab_v3_contrast's DSN head feeds nothing else; the OCR aux head also feeds the
        object-context gather, so it is always in the graph."""
        if not (self.configer.exists('network', 'model_name') and self.configer.exists('loss', 'loss_type')):
            return False
        model = self.configer.get('network', 'model_name')
        loss = self.configer.get('loss', 'loss_type')
        return model.startswith('deeplab_v3') and 'aux' not in loss

    def load_net(self, net):
        net = self.to_device(net)
        net.float()
        resume = self.configer.get('network', 'resume') if self.configer.exists('network', 'resume') else None
        if resume is not None:
            Log.info('Loading checkpoint from {}...'.format(resume))
            blob = torch.load(resume, map_location='cpu')
            if 'state_dict' in blob:
                state = blob['state_dict']
            elif 'model' in blob:
                state = blob['model']
            elif isinstance(blob, OrderedDict):
                state = blob
            else:
                raise RuntimeError('No state_dict found in checkpoint file {}'.format(resume))
            if list(state.keys())[0].startswith('module.'):
                state = {k[7:]: v for k, v in state.items()}
            strict = self.configer.get('network', 'resume_strict') if self.configer.exists('network', 'resume_strict') \
                else False
            self.load_state_dict(net, state, strict)
            if self.configer.exists('network', 'resume_continue') and self.configer.get('network', 'resume_continue'):
                self.configer.update(['network', 'resume'], None)
        return self._make_parallel(net)

    @staticmethod
    def load_state_dict(module, state_dict, strict=False):
        """Non-strict copy with shape check and a report of unexpected / missing keys (reference :121-166)."""
        own = module.state_dict()
        unexpected, mismatched = [], []
        for name, param in state_dict.items():
            if name not in own:
                unexpected.append(name)
                continue
            if own[name].shape != param.shape:
                mismatched.append('{}: {} vs {}'.format(name, tuple(own[name].shape), tuple(param.shape)))
                continue
            own[name].copy_(param)
        from contrastiveseg_amd import kernels as K
        K.SPLIT_WEIGHTS.invalidate()          # packed convolution weights follow the new values (copy_ bumps the version too)
        missing = sorted(set(own.keys()) - set(state_dict.keys()))
        msg = []
        if unexpected:
            msg.append('unexpected key in source state_dict: {}'.format(', '.join(unexpected)))
        if missing:
            msg.append('missing keys in source state_dict: {}'.format(', '.join(missing)))
        if mismatched:
            msg.append('size mismatch: {}'.format('; '.join(mismatched)))
        if msg:
            if strict:
                raise RuntimeError('\n'.join(msg))
            Log.warn('\n'.join(msg))

    def save_net(self, net, save_mode='iters', experiment=None):
        if is_distributed() and get_rank() != 0:
            return
        state = {'config_dict': self.configer.to_dict(), 'state_dict': net.state_dict()}
        ck = self.configer.get('checkpoints')
        root = ck.get('checkpoints_root') or (self.configer.get('project_dir') if self.configer.exists('project_dir')
                                              else '.')
        directory = os.path.join(root, ck['checkpoints_dir'])
        os.makedirs(directory, exist_ok=True)
        name = ck['checkpoints_name']
        torch.save(state, os.path.join(directory, '{}_latest.pth'.format(name)))
        c = self.configer
        if save_mode == 'performance':
            if c.get('performance') > c.get('max_performance'):
                torch.save(state, os.path.join(directory, '{}_max_performance.pth'.format(name)))
                c.update(['max_performance'], c.get('performance'))
        elif save_mode == 'val_loss':
            if c.get('val_loss') < c.get('min_val_loss'):
                torch.save(state, os.path.join(directory, '{}_min_loss.pth'.format(name)))
                c.update(['min_val_loss'], c.get('val_loss'))
        elif save_mode == 'iters':
            if c.get('iters') - c.get('last_iters') >= ck['save_iters']:
                torch.save(state, os.path.join(directory, '{}_iters{}.pth'.format(name, c.get('iters'))))
                c.update(['last_iters'], c.get('iters'))
        elif save_mode == 'epoch':
            if c.get('epoch') - c.get('last_epoch') >= ck['save_epoch']:
                torch.save(state, os.path.join(directory, '{}_epoch{}.pth'.format(name, c.get('epoch'))))
                c.update(['last_epoch'], c.get('epoch'))
        else:
            Log.error('Metric: {} is invalid.'.format(save_mode))
            exit(1)

    def freeze_bn(self, net, syncbn=False):
        for m in net.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d, nn.SyncBatchNorm)):
                m.eval()

    def get_lr(self, optimizer):
        return [g['lr'] for g in optimizer.param_groups]

    def warm_lr(self, iters, scheduler, optimizer, backbone_list=(0,)):
        """reference :271-289"""
        if not self.configer.exists('lr', 'is_warm') or not self.configer.get('lr', 'is_warm'):
            return
        warm = self.configer.get('lr', 'warm')
        if iters < warm['warm_iters']:
            if warm['freeze_backbone']:
                for i in backbone_list:
                    optimizer.param_groups[i]['lr'] = 0.0
            else:
                ratio = (self.configer.get('iters') + 1) / warm['warm_iters']
                base = scheduler.get_lr()
                for i in backbone_list:
                    optimizer.param_groups[i]['lr'] = base[i] * (ratio ** 4)
