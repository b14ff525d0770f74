"""Batch-dict contract of the reference's loaders (lib/datasets/tools/collate.py:37-175; DataHelper.prepare_data,
segmentor/tools/data_helper.py:119-150): {'img': F32 [B,3,H,W], 'labelmap': I64 [B,H,W] (ignore = -1), ...}.
File-based datasets and cv2 augmentation are outside the hot path; `SyntheticLoader` produces seeded batches of the
configured shape that are already resident in HBM, which is what BASELINE.json's throughput metric is quoted on."""
import torch

from contrastiveseg_amd.lib.utils.distributed import get_rank, get_world_size


class DataHelper(object):
    def __init__(self, configer, trainer):
        self.configer = configer
        self.trainer = trainer

    def prepare_data(self, data_dict, want_reverse=False):
        img = data_dict['img']
        target = data_dict['labelmap']
        dev = self.trainer.module_runner.device()
        if img.device != dev:
            img = img.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
        return ([img], target), img.shape[0]


class SyntheticLoader(object):
    """Iterable of `length` identical-shape batches. Per-rank batch = train.batch_size // world_size, as
    lib/datasets/data_loader.py:137 does. mode 'uniform': labels uniform in [-1, K);
    mode 'blocky': rectangles, >= a dozen classes per image."""

    def __init__(self, configer, device, length, seed=304, mode='uniform', fixed=True):
        self.device = device
        self.length = length
        self.fixed = fixed
        self.K = configer.get('data', 'num_classes')
        W, H = configer.get('train', 'data_transformer')['input_size']
        self.B = max(1, configer.get('train', 'batch_size') // get_world_size())
        self.H, self.W = H, W
        self.mode = mode
        self.gen = torch.Generator(device='cpu')
        self.gen.manual_seed(seed + 1000 * get_rank())
        self._batch = self._make() if fixed else None
        self.sampler = None

    def _make(self):
        g = self.gen
        img = torch.randn(self.B, 3, self.H, self.W, generator=g)
        if self.mode == 'uniform':
            lab = torch.randint(-1, self.K, (self.B, self.H, self.W), generator=g)
        else:
            lab = torch.full((self.B, self.H, self.W), -1, dtype=torch.long)
            for b in range(self.B):
                lab[b] = int(torch.randint(0, self.K, (1,), generator=g))
                for _ in range(24):
                    c = int(torch.randint(-1, self.K, (1,), generator=g))
                    y0 = int(torch.randint(0, self.H, (1,), generator=g))
                    x0 = int(torch.randint(0, self.W, (1,), generator=g))
                    hh = int(torch.randint(self.H // 8, self.H // 2 + 1, (1,), generator=g))
                    ww = int(torch.randint(self.W // 8, self.W // 2 + 1, (1,), generator=g))
                    lab[b, y0:y0 + hh, x0:x0 + ww] = c
        return {'img': img.to(self.device), 'labelmap': lab.to(self.device)}

    def __len__(self):
        return self.length

    def __iter__(self):
        for _ in range(self.length):
            yield self._batch if self.fixed else self._make()
