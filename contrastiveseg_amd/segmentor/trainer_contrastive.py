"""Contrastive trainer with the reference's surface (segmentor/trainer_contrastive.py:25-439): `Trainer(configer)`,
`.train()`, the warm-up gate `iters >= contrast.warmup_iters`, the two learning-rate groups, the five wall-clock
meters and the log line, and `_dequeue_and_enqueue` for the memory-bank model.

What changes underneath (MI355X-first):
  * one process per GPU; model wrapped by ModuleRunner in DDP over RCCL/xGMI (no DataParallel path);
  * the criterion is built from HIP kernels (lib/loss/*), with one host sync per step (the per-class counts that
    the CPU randperm stream needs) instead of O(100) implicit syncs of the reference's Python loops;
  * `_dequeue_and_enqueue` is one histogram + one all-class reduction + two row writers on the device; pointer
    arithmetic, last-writer-wins resolution and the CPU `torch.randperm` draws (same order as the reference) stay
    on the host;
  * data: any iterable yielding {'img', 'labelmap'} dicts; by default a seeded synthetic loader resident in HBM
    (file datasets / cv2 augmentation are out of scope)."""
import os
import time
import warnings

import numpy as np
import torch
import torch.nn as nn

from contrastiveseg_amd import _host
from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.loss.loss_manager import LossManager
from contrastiveseg_amd.lib.metrics.running_score import RunningScore
from contrastiveseg_amd.lib.models.model_manager import ModelManager
from contrastiveseg_amd.lib.utils.distributed import (all_gather_cat, exercise_single_rank, get_rank, get_world_size,
                                                      is_distributed)
from contrastiveseg_amd.lib.utils.tools.average_meter import AverageMeter
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log
from contrastiveseg_amd.segmentor.tools.data_helper import DataHelper, SyntheticLoader
from contrastiveseg_amd.segmentor.tools.module_runner import ModuleRunner
from contrastiveseg_amd.segmentor.tools.optim_scheduler import OptimScheduler


# the reference steps the scheduler with an explicit iteration index before optimizer.step() (:193-196); torch warns
# about both habits on every call
warnings.filterwarnings("ignore", message=r"The epoch parameter in `scheduler.step\(\)`")
warnings.filterwarnings("ignore", message=r"Detected call of `lr_scheduler.step\(\)` before `optimizer.step\(\)`")


def _unwrap(net):
    return net.module if hasattr(net, 'module') else net


def plan_enqueue(counts, seg_ptr, pix_ptr, memory_size, pixel_update_freq):
    """Host half of _dequeue_and_enqueue (reference :110-138) on the strided-label histogram `counts` [B,K].
    Walks (image, class>0) in the reference's order, advances the pointers with its arithmetic (segment pointer +1;
    pixel pointer +1 -- not +K -- or reset to 0 on wrap) and draws torch.randperm(num_pixel) per pair from the CPU
    generator. Later writes to the same bank row win. Returns (segment jobs, pixel rows, new pointers)."""
    B, Kc = counts.shape
    seg_ptr = seg_ptr.copy()
    pix_ptr = pix_ptr.copy()
    seg_last, pix_last = {}, {}
    pairs = [(bs, lb, int(counts[bs, lb])) for bs in range(B) for lb in range(1, Kc) if counts[bs, lb] > 0]
    # torch.randperm(num_pixel) per (image, class) in the reference's order (:127), one native call
    perms = _host.randperm_prefixes([n for _, _, n in pairs], [min(n, pixel_update_freq) for _, _, n in pairs])
    for (bs, lb, n), perm in zip(pairs, perms):
        seg_last[(lb, int(seg_ptr[lb]))] = bs
        seg_ptr[lb] = (seg_ptr[lb] + 1) % memory_size
        k = min(n, pixel_update_freq)
        ptr = int(pix_ptr[lb])
        if ptr + k >= memory_size:
            rows = range(memory_size - k, memory_size)
            pix_ptr[lb] = 0
        else:
            rows = range(ptr, ptr + k)
            pix_ptr[lb] = (pix_ptr[lb] + 1) % memory_size
        for r, pos in zip(rows, perm):
            pix_last[(lb, r)] = (bs, int(pos))
    seg_jobs = np.array([(b, lb, row) for (lb, row), b in seg_last.items()], dtype=np.int32).reshape(-1, 3)
    pix_rows = np.array([(b, pos, lb, row) for (lb, row), (b, pos) in pix_last.items()], dtype=np.int32).reshape(-1, 4)
    return seg_jobs, pix_rows, seg_ptr, pix_ptr


class Trainer(object):
    def __init__(self, configer, train_loader=None, val_loader=None):
        self.configer = configer
        self.batch_time = AverageMeter()
        self.foward_time = AverageMeter()
        self.backward_time = AverageMeter()
        self.loss_time = AverageMeter()
        self.data_time = AverageMeter()
        self.train_losses = AverageMeter()
        self.val_losses = AverageMeter()
        self.loss_manager = LossManager(configer)
        self.module_runner = ModuleRunner(configer)
        self.model_manager = ModelManager(configer)
        self.optim_scheduler = OptimScheduler(configer)
        self.data_helper = DataHelper(configer, self)
        self.seg_net = None
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.optimizer = None
        self.scheduler = None
        self._init_model()

    def _init_model(self):
        self.seg_net = self.model_manager.semantic_segmentor()
        if self.configer.exists('network', 'channels_last') and self.configer.get('network', 'channels_last'):
            self.seg_net = self.seg_net.to(memory_format=torch.channels_last)
        self.seg_net = self.module_runner.load_net(self.seg_net)
        # single-rank GPU runs: the segmentor's forward and backward are replayed as two hipGraphs (segmentor/tools/step_graph.py);
        # everything the graphs do not cover (eval, other shapes, multi-rank) stays on the eager path
        from contrastiveseg_amd.segmentor.tools import step_graph
        self.step_graph = step_graph.install(_unwrap(self.seg_net), self.configer)

        Log.info('Params Group Method: {}'.format(self.configer.get('optim', 'group_method')))
        if self.configer.get('optim', 'group_method') == 'decay':
            params_group = self.group_weight(self.seg_net)
        else:
            assert self.configer.get('optim', 'group_method') is None
            params_group = self._get_parameters()
        self.optimizer, self.scheduler = self.optim_scheduler.init_optimizer(params_group)

        if self.train_loader is None:
            data_dir = self.configer.get('data', 'data_dir') if self.configer.exists('data', 'data_dir') else None
            if isinstance(data_dir, (list, tuple)):
                data_dir = data_dir[0] if data_dir else None
            if data_dir and os.path.isdir(os.path.join(data_dir, 'train', 'image')):
                # files on disk: host decodes, the GPU augments / normalises / collates (lib/datasets/data_loader.py)
                from contrastiveseg_amd.lib.datasets.data_loader import DataLoader
                self.configer.update(['data', 'data_dir'], data_dir)
                loader = DataLoader(self.configer, self.module_runner.device())
                self.train_loader = loader.get_trainloader()
                if self.val_loader is None and os.path.isdir(os.path.join(data_dir, 'val', 'image')) \
                        and self.configer.exists('val', 'data_transformer'):
                    self.val_loader = loader.get_valloader()
            else:
                self.train_loader = SyntheticLoader(self.configer, self.module_runner.device(),
                                                    length=self.configer.get('solver', 'max_iters'))
        self.pixel_loss = self.module_runner.to_device(self.loss_manager.get_seg_loss())

        self.with_contrast = True if self.configer.exists("contrast") else False
        self.contrast_warmup_iters = self.configer.get("contrast", "warmup_iters") \
            if self.configer.exists("contrast", "warmup_iters") else 0
        self.with_memory = self.configer.exists('contrast', 'with_memory')
        if self.with_memory:
            self.memory_size = self.configer.get('contrast', 'memory_size')
            self.pixel_update_freq = self.configer.get('contrast', 'pixel_update_freq')
        self.network_stride = self.configer.get('network', 'stride')
        Log.info("with_contrast: {}, warmup_iters: {}, with_memory: {}".format(
            self.with_contrast, self.contrast_warmup_iters, self.with_memory))

    # ------------------------------------------------------------------------------------------------------
    def _dequeue_and_enqueue(self, keys, labels, segment_queue, segment_queue_ptr, pixel_queue, pixel_queue_ptr):
        """reference :102-138; all four queue tensors are updated in place. Multi-rank runs: _enqueue_global (the same update from
        the GLOBAL batch on every rank)."""
        if get_world_size() > 1 or exercise_single_rank():
            return self._enqueue_global(keys, labels, segment_queue, segment_queue_ptr, pixel_queue, pixel_queue_ptr)
        Kc = segment_queue.shape[0]
        keys = keys.contiguous()
        counts_d = K.queue_count(labels, self.network_stride, Kc)
        sums = K.queue_class_sums(keys, labels, self.network_stride, Kc)     # queued before the host sync below
        host = torch.cat([counts_d.reshape(-1).long(), segment_queue_ptr, pixel_queue_ptr]).cpu().numpy()
        B = keys.shape[0]
        counts = host[:B * Kc].reshape(B, Kc)
        seg_ptr, pix_ptr = host[B * Kc:B * Kc + Kc], host[B * Kc + Kc:]
        seg_jobs, pix_rows, seg_ptr, pix_ptr = plan_enqueue(counts, seg_ptr, pix_ptr, self.memory_size,
                                                            self.pixel_update_freq)
        dev = keys.device
        if len(seg_jobs):
            j = torch.from_numpy(seg_jobs).to(dev)
            K.queue_write_segments(sums, counts_d, j[:, 0].contiguous(), j[:, 1].contiguous(), j[:, 2].contiguous(),
                                   segment_queue)
        if len(pix_rows):
            r = torch.from_numpy(pix_rows).to(dev)
            K.queue_write_pixels(keys, r[:, 0].contiguous(), r[:, 1].contiguous(), r[:, 2].contiguous(),
                                 r[:, 3].contiguous(), pixel_queue)
        segment_queue_ptr.copy_(torch.from_numpy(seg_ptr.astype(np.int64)))
        pixel_queue_ptr.copy_(torch.from_numpy(pix_ptr.astype(np.int64)))

    def _enqueue_global(self, keys, labels, segment_queue, segment_queue_ptr, pixel_queue, pixel_queue_ptr):
        """The memory bank under data parallelism (exchange 4). The reference lets every rank enqueue its own
        images and then has DDP broadcast rank 0's queues over everybody else's before each forward (module_runner.py:62-76
        `broadcast_buffers`; at memory_size 5000 that is 2 x 97 MB per step, and the other ranks' keys are lost). Here every rank
        applies the update of the GLOBAL batch -- images in rank order = the batch a single process would see -- so the banks stay
        identical without any buffer broadcast:
          1. all-gather of the per-(image, class) strided-label counts and class sums ([B, K, D + 1] floats per rank);
          2. every rank walks the global (image, class) pairs with plan_enqueue (pointer arithmetic is a function of the counts;
             the torch.randperm draws are rank 0's -- one broadcast of the drawn positions -- so ranks whose CPU generators have
             diverged still agree, and rank 0 draws exactly what a single process on the concatenated batch draws);
          3. each rank reads the selected pixels of ITS images; one all-reduce assembles the [rows, D] selection (each row has
             one owner, the others contribute zeros);
          4. identical writes on every rank."""
        import torch.distributed as dist
        world, rank = get_world_size(), get_rank()
        Kc, ms, Dq = segment_queue.shape
        keys = keys.contiguous()
        B, D = keys.shape[:2]
        dev = keys.device
        counts_l = K.queue_count(labels, self.network_stride, Kc)
        sums_l = K.queue_class_sums(keys, labels, self.network_stride, Kc)
        packed = all_gather_cat(torch.cat([sums_l, counts_l.to(sums_l.dtype).unsqueeze(2)], dim=2))        # [world * B, K, D + 1]
        counts_d = packed[:, :, D].round().to(torch.int32).contiguous()
        sums = packed[:, :, :D].contiguous()
        host = torch.cat([counts_d.reshape(-1).long(), segment_queue_ptr, pixel_queue_ptr]).cpu().numpy()
        Bg = world * B
        counts = host[:Bg * Kc].reshape(Bg, Kc)
        seg_ptr, pix_ptr = host[Bg * Kc:Bg * Kc + Kc], host[Bg * Kc + Kc:]
        seg_jobs, pix_rows, seg_ptr, pix_ptr = plan_enqueue(counts, seg_ptr, pix_ptr, self.memory_size, self.pixel_update_freq)
        if len(seg_jobs):
            j = torch.from_numpy(seg_jobs).to(dev)
            K.queue_write_segments(sums, counts_d, j[:, 0].contiguous(), j[:, 1].contiguous(), j[:, 2].contiguous(), segment_queue)
        if len(pix_rows):
            on_dev = dist.get_backend() == "nccl"
            pos = torch.from_numpy(np.ascontiguousarray(pix_rows[:, 1]).astype(np.int64))
            pos = pos.to(dev) if on_dev else pos
            dist.broadcast(pos, src=0)                       # rank 0's draws decide
            pos = pos.to(dev)
            r = torch.from_numpy(pix_rows).to(dev).long()
            owner = torch.div(r[:, 0], B, rounding_mode='floor')
            mine = torch.nonzero(owner == rank).reshape(-1)
            rows = torch.zeros(len(pix_rows), D, dtype=keys.dtype, device=dev)
            if mine.numel():
                rows[mine] = keys.view(B, D, -1)[r[mine, 0] - rank * B, :, pos[mine]]
            dist.all_reduce(rows)                            # every row has exactly one owner: x + 0 + ... + 0 is exact
            pixel_queue[r[:, 2], r[:, 3]] = nn.functional.normalize(rows, p=2, dim=1)      # targets are unique (last writer wins on the host)
        segment_queue_ptr.copy_(torch.from_numpy(seg_ptr.astype(np.int64)))
        pixel_queue_ptr.copy_(torch.from_numpy(pix_ptr.astype(np.int64)))

    @staticmethod
    def group_weight(module):
        """reference :141-161"""
        group_decay, group_no_decay = [], []
        for m in module.modules():
            if isinstance(m, (nn.Linear, nn.modules.conv._ConvNd)):
                group_decay.append(m.weight)
                if m.bias is not None:
                    group_no_decay.append(m.bias)
            else:
                if hasattr(m, 'weight') and isinstance(getattr(m, 'weight'), nn.Parameter):
                    group_no_decay.append(m.weight)
                if hasattr(m, 'bias') and isinstance(getattr(m, 'bias'), nn.Parameter):
                    group_no_decay.append(m.bias)
        assert len(list(module.parameters())) == len(group_decay) + len(group_no_decay)
        return [dict(params=group_decay), dict(params=group_no_decay, weight_decay=.0)]

    def _get_parameters(self):
        """reference :163-175: backbone lr / head lr * nbb_mult"""
        bb_lr, nbb_lr = [], []
        for key, value in dict(self.seg_net.named_parameters()).items():
            (nbb_lr if 'backbone' not in key else bb_lr).append(value)
        base = self.configer.get('lr', 'base_lr')
        return [{'params': bb_lr, 'lr': base}, {'params': nbb_lr, 'lr': base * self.configer.get('lr', 'nbb_mult')}]

    # ------------------------------------------------------------------------------------------------------
    def train_step(self, data_dict):
        """One iteration of the reference's loop body (:193-267). Returns the (rank-local) loss tensor."""
        c = self.configer
        start_time = getattr(self, '_t_last', time.time())
        if c.get('lr', 'metric') == 'iters':
            self.scheduler.step(c.get('iters'))
        else:
            self.scheduler.step(c.get('epoch'))
        if c.exists('lr', 'is_warm') and c.get('lr', 'is_warm'):
            self.module_runner.warm_lr(c.get('iters'), self.scheduler, self.optimizer, backbone_list=[0, ])

        (inputs, targets), batch_size = self.data_helper.prepare_data(data_dict)
        self.data_time.update(time.time() - start_time)

        t0 = time.time()
        with_embed = True if c.get('iters') >= self.contrast_warmup_iters else False
        net = _unwrap(self.seg_net)
        if self.with_contrast and self.with_memory:
            outputs = self.seg_net(*inputs, targets, with_embed=with_embed)
            outputs['pixel_queue'] = net.pixel_queue
            outputs['pixel_queue_ptr'] = net.pixel_queue_ptr
            outputs['segment_queue'] = net.segment_queue
            outputs['segment_queue_ptr'] = net.segment_queue_ptr
        elif self.with_contrast:
            outputs = self.seg_net(*inputs, with_embed=with_embed)
        else:
            outputs = self.seg_net(*inputs)
        self.foward_time.update(time.time() - t0)

        t0 = time.time()
        loss = self.pixel_loss(outputs, targets, with_embed=with_embed)
        self.loss_time.update(time.time() - t0)

        t0 = time.time()
        self.optimizer.zero_grad(set_to_none=True)
        with K.wgrad_scope(loss.device):          # weight gradients on a side stream, joined before the optimizer (kernels.py)
            loss.backward()
        if self.with_memory and 'key' in outputs and 'lb_key' in outputs:
            # The reference enqueues between the loss and backward (:246-251); its loss holds a torch.cat COPY of the
            # bank, so its gradient is that of the bank as it was during the forward. Here cseg_contrast_bwd re-reads
            # the bank rows in place, so the in-place enqueue must come after backward: the keys are detached, the
            # RNG draws are the same, hence bank contents, pointers and gradients are identical to the reference's.
            # (kernels.PixelContrast.backward raises if the bank was modified in between.)
            self._dequeue_and_enqueue(outputs['key'], outputs['lb_key'], segment_queue=net.segment_queue,
                                      segment_queue_ptr=net.segment_queue_ptr, pixel_queue=net.pixel_queue,
                                      pixel_queue_ptr=net.pixel_queue_ptr)
        self.optimizer.step()
        self.backward_time.update(time.time() - t0)

        self._last_loss = loss.detach()
        self._last_batch = batch_size
        self.batch_time.update(time.time() - start_time)
        self._t_last = time.time()
        c.plus_one('iters')
        return self._last_loss

    def _display(self):
        """The reference reduces the loss to rank 0 and calls .item() every step (:228-254); that is a host sync per
        step, so here it happens only when the line is printed."""
        c = self.configer
        bad = sum((m.status[1] for m in self.pixel_loss.modules() if hasattr(m, 'bad_label_count')),
                  torch.zeros((), dtype=torch.int32, device=self._last_loss.device))
        disp = torch.stack([self._last_loss.float(), bad.float()])
        if is_distributed() and get_world_size() > 1:
            import torch.distributed as dist
            dist.all_reduce(disp)               # every rank learns about bad labels on any rank
            disp[0] = disp[0] / get_world_size()
        disp = disp.tolist()
        if disp[1] > 0:
            raise RuntimeError("%d label values were neither ce_ignore_index nor in [0, num_classes): the fused CE "
                               "kernel dropped them (nn.CrossEntropyLoss would have asserted); fix the label ids or "
                               "loss.params.ce_ignore_index" % int(disp[1]))
        self.train_losses.update(disp[0], self._last_batch)
        if not is_distributed() or get_rank() == 0:
            Log.info('Train Epoch: {0}\tTrain Iteration: {1}\t'
                     'Time {batch_time.sum:.3f}s / {2}iters, ({batch_time.avg:.3f})\t'
                     'Forward Time {foward_time.sum:.3f}s / {2}iters, ({foward_time.avg:.3f})\t'
                     'Backward Time {backward_time.sum:.3f}s / {2}iters, ({backward_time.avg:.3f})\t'
                     'Loss Time {loss_time.sum:.3f}s / {2}iters, ({loss_time.avg:.3f})\t'
                     'Data load {data_time.sum:.3f}s / {2}iters, ({data_time.avg:3f})\n'
                     'Learning rate = {3}\tLoss = {loss.val:.8f} (ave = {loss.avg:.8f})\n'.format(
                         c.get('epoch'), c.get('iters'), c.get('solver', 'display_iter'),
                         self.module_runner.get_lr(self.optimizer), batch_time=self.batch_time,
                         foward_time=self.foward_time, backward_time=self.backward_time, loss_time=self.loss_time,
                         data_time=self.data_time, loss=self.train_losses))
        for m in (self.batch_time, self.foward_time, self.backward_time, self.loss_time, self.data_time,
                  self.train_losses):
            m.reset()

    def __train(self):
        self.seg_net.train()
        self.pixel_loss.train()
        c = self.configer
        self._t_last = time.time()
        if hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):
            self.train_loader.sampler.set_epoch(c.get('epoch'))
        for data_dict in self.train_loader:
            self.train_step(data_dict)
            if c.get('iters') % c.get('solver', 'display_iter') == 0:
                self._display()
            if c.get('iters') == c.get('solver', 'max_iters'):
                break
            if self.val_loader is not None and c.get('iters') % c.get('solver', 'test_interval') == 0:
                self.__val()
        c.plus_one('epoch')

    @torch.no_grad()
    def __val(self, data_loader=None):
        """Validation loss + mIoU (reference :306-401 -> StandardEvaluator -> RunningScore). Logits are upsampled to the
        label size (bilinear, align_corners=True -- the reference's evaluator resizes with cv2 INTER_CUBIC on the host,
        which is outside the hot path), arg-maxed, and accumulated into an on-device confusion matrix with the
        reference's RunningScore arithmetic (lib/metrics/running_score.py of this package)."""
        self.seg_net.eval()
        self.pixel_loss.eval()
        score = RunningScore(self.configer, ignore_index=-1)
        for data_dict in (self.val_loader if data_loader is None else data_loader):
            (inputs, targets), batch_size = self.data_helper.prepare_data(data_dict)
            outputs = self.seg_net(*inputs, is_eval=True)
            self.val_losses.update(self.pixel_loss(outputs, targets).item(), batch_size)
            seg = nn.functional.interpolate(outputs['seg'], size=targets.shape[-2:], mode='bilinear',
                                            align_corners=True)
            score.update(seg.argmax(1), targets)
        # every rank joins the all-reduce, whether or not its validation shard held a batch (a rank that skipped it would
        # leave the others waiting); "was anything validated" is then read off the REDUCED matrix, identically on all ranks
        score.reduce_scores()
        seen = float(score.reduced_confusion_matrix.sum()) > 0
        if seen:
            miou = float(score.get_mean_iou())
            self.last_val_score = score
            self.configer.update(['performance'], miou)
            self.configer.update(['val_loss'], self.val_losses.avg)
            if self.configer.exists('checkpoints'):
                self.module_runner.save_net(self.seg_net, save_mode='performance')
                self.module_runner.save_net(self.seg_net, save_mode='val_loss')
            Log.info('Val mIoU {:.4f}\tPixel acc {:.4f}\tLoss {:.8f}'.format(miou, float(score.get_pixel_acc()),
                                                                            self.val_losses.avg))
        self.val_losses.reset()
        self.seg_net.train()
        self.pixel_loss.train()

    def validate(self, data_loader=None):
        """Public entry to the validation pass (the reference calls the name-mangled __val from train() only)."""
        return self.__val(data_loader)

    def train(self):
        """reference :403-427 (SWA tail omitted: torchcontrib is outside the hot path)."""
        c = self.configer
        if c.exists('network', 'resume') and c.get('network', 'resume') is not None and c.exists('network', 'resume_val') \
                and c.get('network', 'resume_val') and self.val_loader is not None:
            self.__val()
        while c.get('iters') < c.get('solver', 'max_iters'):
            self.__train()
        if self.val_loader is not None:
            self.__val()
