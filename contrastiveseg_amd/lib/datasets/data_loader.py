"""Train / validation loaders with the reference's entry points (lib/datasets/data_loader.py:27-140: `DataLoader(configer)
.get_trainloader() / .get_valloader()`) and directory layout (lib/datasets/loader/default_loader.py:108-200:
`<data_dir>/<split>/image/*.png|jpg` + `<data_dir>/<split>/label/<same stem>.png`).

MI355X-first split of the work: the host only DECODES files (PIL, a small thread pool) and hands
raw uint8 batches to the device through pinned memory on a side HIP stream, one batch ahead of the training step;
augmentation, tensor conversion, label encoding and collation are one kernel on the GPU
(lib/datasets/tools/gpu_aug.py -> cseg_augment_batch). Sample order: a fresh torch.randperm per epoch like the
reference's RandomSampler; with a process group every rank takes its strided share of it (DistributedSampler semantics,
data_loader.py:137: batch_size // world_size per rank). Images of one batch must share a size (Cityscapes does;
mixed-size datasets are outside the accelerated path)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from contrastiveseg_amd.lib.datasets.tools.gpu_aug import GPUBatchTransform
from contrastiveseg_amd.lib.utils.distributed import get_rank, get_world_size
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

IMG_EXT = ('.png', '.jpg', '.jpeg', '.bmp')


def list_pairs(root_dir, split):
    """(image path, label path) pairs of default_loader.py:108-200's layout, sorted by name."""
    image_dir = os.path.join(root_dir, split, 'image')
    label_dir = os.path.join(root_dir, split, 'label')
    pairs = []
    for name in sorted(os.listdir(image_dir)):
        stem, ext = os.path.splitext(name)
        if ext.lower() not in IMG_EXT:
            continue
        lab = os.path.join(label_dir, stem + '.png')
        if not os.path.exists(lab):
            Log.error('Label Path: {} not exists.'.format(lab))
            continue
        pairs.append((os.path.join(image_dir, name), lab))
    return pairs


def _decode(pair, bgr):
    from PIL import Image
    img = np.asarray(Image.open(pair[0]).convert('RGB'))
    if bgr:
        img = img[:, :, ::-1]
    lab = np.asarray(Image.open(pair[1]))
    if lab.ndim == 3:
        lab = lab[:, :, 0]
    return np.ascontiguousarray(img), np.ascontiguousarray(lab.astype(np.uint8))


class FolderSource(object):
    """Raw uint8 batches from image / label files."""

    def __init__(self, configer, split, batch_size, shuffle, drop_last=True, workers=8):
        self.pairs = list_pairs(configer.get('data', 'data_dir'), split)
        if not self.pairs:
            raise RuntimeError('no image/label pairs under {}/{}'.format(configer.get('data', 'data_dir'), split))
        mode = configer.get('data', 'input_mode') if configer.exists('data', 'input_mode') else 'BGR'
        self.bgr = (mode == 'BGR')
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.epoch = 0
        self.seed = configer.get('seed') if configer.exists('seed') and configer.get('seed') is not None else 0

    def set_epoch(self, epoch):
        """torch.utils.data.DistributedSampler.set_epoch: the permutation of an epoch is a function of (seed, epoch) only."""
        self.epoch = int(epoch)

    def shard(self, epoch=None):
        """This rank's sample indices for one epoch, the way the reference's DistributedSampler (lib/datasets/data_loader.py:81-82
        of the reference -> torch.utils.data.distributed.DistributedSampler) produces them: a permutation drawn from a PRIVATE
        generator seeded with seed + epoch (identical on every rank, untouched by the randperm calls of the anchor sampling / the
        memory bank, which advance the global CPU generator by rank-dependent amounts), cut to a multiple of world x batch when
        drop_last (train), padded by wrapping around otherwise, so that every rank yields the same number of batches (a rank with
        fewer forward calls would hang DDP's collectives), then strided by rank."""
        n, world, rank = len(self.pairs), get_world_size(), get_rank()
        epoch = self.epoch if epoch is None else epoch
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(int(self.seed) + int(epoch))
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        chunk = world * self.batch_size
        if self.drop_last:
            order = order[:(n // chunk) * chunk]
        else:
            total = ((n + world - 1) // world) * world
            order = order + order[:total - n]              # n >= 1, total - n < world <= ... wraps once at most for n >= world
            while len(order) < total:
                order = order + order[:total - len(order)]
        return order[rank::world]

    def __len__(self):
        n, world = len(self.pairs), get_world_size()
        if self.drop_last:
            return n // (world * self.batch_size)
        per_rank = (n + world - 1) // world
        return (per_rank + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.shard()
        self.epoch += 1                                    # a caller that never calls set_epoch still gets a new permutation
        for i in range(0, len(order), self.batch_size):
            items = list(self.pool.map(lambda k: _decode(self.pairs[k], self.bgr), order[i:i + self.batch_size]))
            if len({it[0].shape for it in items}) != 1:
                raise RuntimeError('images of one batch differ in size; the accelerated loader needs a common size')
            yield np.stack([it[0] for it in items]), np.stack([it[1] for it in items])


class SyntheticRawSource(object):
    """Seeded raw uint8 images + label ids of `data.raw_size` (default: twice the input size), for runs without files."""

    def __init__(self, configer, batch_size, length, seed=304):
        W, H = configer.get('train', 'data_transformer')['input_size']
        if configer.exists('data', 'raw_size'):
            W0, H0 = configer.get('data', 'raw_size')
        else:
            W0, H0 = 2 * W, 2 * H
        self.shape, self.batch_size, self.length = (H0, W0), batch_size, length
        self.ids = configer.get('data', 'label_list') if configer.exists('data', 'label_list') else \
            list(range(configer.get('data', 'num_classes')))
        self.rs = np.random.RandomState(seed + 1000 * get_rank())

    def __len__(self):
        return self.length

    def __iter__(self):
        H0, W0 = self.shape
        for _ in range(self.length):
            img = self.rs.randint(0, 256, size=(self.batch_size, H0, W0, 3)).astype(np.uint8)
            lab = np.full((self.batch_size, H0, W0), 255, np.uint8)
            for b in range(self.batch_size):
                for _r in range(12):
                    y0, x0 = self.rs.randint(0, H0), self.rs.randint(0, W0)
                    lab[b, y0:y0 + H0 // 3, x0:x0 + W0 // 3] = self.ids[self.rs.randint(0, len(self.ids))]
            yield img, lab


class GPUAugLoader(object):
    """Iterable of {'img': f32 [B,3,H,W], 'labelmap': i64 [B,H,W]} batches on `device`. Raw batch k+1 is decoded and
    uploaded (pinned memory, side stream) while the caller trains on batch k."""

    def __init__(self, configer, source, device, split='train'):
        self.source, self.device = source, device
        self.transform = GPUBatchTransform(configer, split)
        # the trainer calls `loader.sampler.set_epoch(epoch)` like the reference does (trainer_contrastive.py:183-184)
        self.sampler = source if hasattr(source, 'set_epoch') else None
        self._stream = None

    def __len__(self):
        return len(self.source)

    def _upload(self, raw):
        img, lab = raw
        if self.device.type != 'cuda':
            return torch.from_numpy(img), torch.from_numpy(lab), None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._stream):
            ti = torch.from_numpy(img).pin_memory().to(self.device, non_blocking=True)
            tl = torch.from_numpy(lab).pin_memory().to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._stream)
        return ti, tl, done

    def __iter__(self):
        it = iter(self.source)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            ti, tl, done = nxt
            try:
                nxt = self._upload(next(it))        # overlaps the kernel + the training step of the current batch
            except StopIteration:
                nxt = None
            if done is not None:
                torch.cuda.current_stream(self.device).wait_event(done)
                ti.record_stream(torch.cuda.current_stream(self.device))
                tl.record_stream(torch.cuda.current_stream(self.device))
            out = self.transform(ti, tl)
            yield {'img': out['img'], 'labelmap': out['labelmap']}


class DataLoader(object):
    def __init__(self, configer, device=None):
        self.configer = configer
        self.device = device if device is not None else torch.device(
            'cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')

    def _workers(self):
        return self.configer.get('data', 'workers') if self.configer.exists('data', 'workers') else 8

    def get_trainloader(self):
        bs = max(1, self.configer.get('train', 'batch_size') // get_world_size())
        src = FolderSource(self.configer, 'train', bs, shuffle=True, drop_last=True, workers=self._workers())
        Log.info('train: {} image/label pairs, {} batches of {} per rank'.format(len(src.pairs), len(src), bs))
        return GPUAugLoader(self.configer, src, self.device, 'train')

    def get_valloader(self, dataset='val'):
        bs = max(1, self.configer.get('val', 'batch_size') // get_world_size())
        src = FolderSource(self.configer, dataset, bs, shuffle=False, drop_last=False, workers=self._workers())
        return GPUAugLoader(self.configer, src, self.device, 'val')
