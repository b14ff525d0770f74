"""Training-time augmentation, tensor conversion and batch collation of the reference, re-designed for the GPU
(SURVEY.md section 8 f4). Same config schema as the reference:

    train_trans.trans_seq = [random_resize, random_crop, random_hflip, random_brightness]   (cv2_aug_transforms.py:654-760)
    normalize = {div_value, mean, std}                                                     (data_loader.py:39-49)
    train.data_transformer = {size_mode: fix_size, input_size: [W, H], align_method: only_pad, pad_mode: random}
    data.label_list (raw id -> train id)                                                    (default_loader.py:94-106)

What moves where: every RANDOM DECISION stays on the host and is drawn from Python's `random` module in exactly the
reference's order (per sample: resize scale, aspect, resize coin; crop x, y, crop coin; flip coin; brightness coin and
shift -- then per batch the collate pad offsets), so a seed gives the same geometric decisions as the reference loader
with workers=0. Everything that touches pixels -- cv2.resize, the crop slice, cv2.flip, the brightness arithmetic,
ToTensor / Normalize, label encoding, ReLabel and the padding of collate -- is one kernel launch per batch on the raw
uint8 images (cseg_augment_batch): the host only decodes files and uploads uint8."""
import math
import random

import numpy as np
import torch

from contrastiveseg_amd import kernels as K

SUPPORTED = ('random_resize', 'random_crop', 'random_hflip', 'random_brightness')


class SampleParams(object):
    """Decisions for one sample, in the reference's drawing order."""
    __slots__ = ('Wr', 'Hr', 'x_off', 'y_off', 'tw', 'th', 'flip', 'shift')

    def as_list(self, left_pad=0, up_pad=0):
        return [self.Wr, self.Hr, self.x_off, self.y_off, self.tw, self.th, int(self.flip), self.shift, left_pad, up_pad,
                0, 0]


class GPUAugCompose(object):
    """Host half of CV2AugCompose (cv2_aug_transforms.py:654-760): parses the same config and draws the decisions."""

    def __init__(self, configer, split='train'):
        self.configer = configer
        self.split = split
        key = 'train_trans' if split == 'train' else 'val_trans'
        self.cfg = configer.get(key) if configer.exists(key) else {'trans_seq': []}
        if self.cfg.get('shuffle_trans_seq'):
            raise NotImplementedError('shuffle_trans_seq is outside the accelerated data path')
        self.seq = list(self.cfg.get('trans_seq', []))
        for name in self.seq:
            if name not in SUPPORTED:
                raise NotImplementedError('augmentation {!r} is outside the accelerated data path; supported: {}'
                                          .format(name, SUPPORTED))
        rr = self.cfg.get('random_resize', {})
        if 'random_resize' in self.seq and rr.get('method', 'random') != 'random':
            raise NotImplementedError("random_resize.method {!r}: only 'random' is implemented".format(rr.get('method')))
        rc = self.cfg.get('random_crop', {})
        if 'random_crop' in self.seq and rc.get('method', 'random') not in ('random', 'center'):
            raise NotImplementedError("random_crop.method {!r}: only 'random' / 'center'".format(rc.get('method')))
        hf = self.cfg.get('random_hflip', {})
        if 'random_hflip' in self.seq and hf.get('swap_pair'):
            raise NotImplementedError('random_hflip.swap_pair (left/right label swap) is not implemented')

    def draw(self, width, height):
        """One sample of size (width, height). Reference order of `random` calls:
        RandomResize.__call__ :405-443 (uniform scale | scale_list index, uniform aspect, coin), RandomCrop :580-603
        (randint x, randint y, coin), RandomHFlip :202-208 (coin), RandomBrightness :318-324 + :310-315 (coin, then
        randint shift only when applied)."""
        p = SampleParams()
        p.Wr, p.Hr = width, height
        p.x_off = p.y_off = 0
        p.tw, p.th = width, height
        p.flip, p.shift = False, 0
        for name in self.seq:
            c = self.cfg[name]
            if name == 'random_resize':
                if c.get('scale_list') is None:
                    scale = random.uniform(c['scale_range'][0], c['scale_range'][1])
                else:
                    scale = c['scale_list'][random.randint(0, len(c['scale_list']) - 1)]
                aspect = random.uniform(*c['aspect_range'])
                w_ratio = math.sqrt(aspect) * scale
                h_ratio = math.sqrt(1.0 / aspect) * scale
                bound = c.get('max_side_bound')
                if bound is not None and max(p.Hr * h_ratio, p.Wr * w_ratio) > bound:
                    d = bound / max(p.Hr * h_ratio, p.Wr * w_ratio)
                    w_ratio *= d
                    h_ratio *= d
                size = (int(p.Wr * w_ratio), int(p.Hr * h_ratio))
                if not (random.random() > c['ratio']):
                    p.Wr, p.Hr = size
                    p.tw, p.th = size
            elif name == 'random_crop':
                cw, ch = c['crop_size']
                tw, th = min(cw, p.tw), min(ch, p.th)
                if c.get('method', 'random') == 'center':
                    x, y = (p.tw - tw) // 2, (p.th - th) // 2
                else:
                    x = random.randint(0, p.tw - tw)
                    y = random.randint(0, p.th - th)
                if not (random.random() > c['ratio']):
                    p.x_off, p.y_off, p.tw, p.th = p.x_off + x, p.y_off + y, tw, th
            elif name == 'random_hflip':
                if not (random.random() > c['ratio']):
                    p.flip = not p.flip
            elif name == 'random_brightness':
                if not (random.random() > c['ratio']):
                    p.shift += random.randint(-c['shift_value'], c['shift_value'])
        if p.Wr <= 0 or p.Hr <= 0:
            raise RuntimeError('random_resize produced an empty image ({}x{})'.format(p.Wr, p.Hr))
        return p


class GPUBatchTransform(object):
    """aug_transform + img_transform + label_transform + collate of the reference's train loader
    (lib/datasets/data_loader.py:130-140 -> DefaultLoader.__getitem__ -> collate) for one batch."""

    def __init__(self, configer, split='train'):
        self.configer = configer
        self.aug = GPUAugCompose(configer, split)
        dt = configer.get('train' if split == 'train' else 'val', 'data_transformer')
        if dt.get('size_mode', 'fix_size') != 'fix_size':
            raise NotImplementedError("size_mode {!r}: only 'fix_size' is implemented".format(dt.get('size_mode')))
        if dt.get('align_method', 'only_pad') != 'only_pad':
            raise NotImplementedError("align_method {!r}: only 'only_pad' is implemented".format(dt.get('align_method')))
        self.pad_mode = dt.get('pad_mode', 'random')
        if self.pad_mode not in ('random', 'pad_left_up', 'pad_right_down', 'pad_center'):
            raise NotImplementedError('pad_mode {!r}'.format(self.pad_mode))
        self.target_w, self.target_h = dt['input_size']
        norm = configer.get('normalize') if configer.exists('normalize') else \
            {'div_value': 255.0, 'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}
        self.div, self.mean, self.std = norm['div_value'], norm['mean'], norm['std']
        self.lut = None
        if configer.exists('data', 'label_list'):
            lut = np.full(256, 255, dtype=np.int16)               # default_loader.py:94-106: unlisted ids -> 255
            for i, cid in enumerate(configer.get('data', 'label_list')):
                lut[cid] = i
            self.lut = torch.from_numpy(lut)
        elif configer.exists('data', 'reduce_zero_label') and configer.get('data', 'reduce_zero_label'):
            lut = np.arange(256, dtype=np.int16) - 1              # default_loader.py:83-92 (uint8 wrap: 0 -> 255)
            lut[0] = 255
            self.lut = torch.from_numpy(lut)

    def plan(self, sizes):
        """-> int32 [B, AUG_PARAM_INTS]. Sample decisions first (one __getitem__ per sample), then the collate pads in
        sample order (collate.py:108-121)."""
        samples = [self.aug.draw(w, h) for (w, h) in sizes]
        rows = []
        for p in samples:
            pad_w, pad_h = self.target_w - p.tw, self.target_h - p.th
            if pad_w < 0 or pad_h < 0:
                raise RuntimeError('sample of {}x{} exceeds the fixed input size {}x{} (the reference asserts here, '
                                   'collate.py:106)'.format(p.tw, p.th, self.target_w, self.target_h))
            left = up = 0
            if pad_w > 0 or pad_h > 0:
                if self.pad_mode == 'random':
                    left = random.randint(0, pad_w)
                    up = random.randint(0, pad_h)
                elif self.pad_mode == 'pad_left_up':
                    left, up = pad_w, pad_h
                elif self.pad_mode == 'pad_center':
                    left, up = pad_w // 2, pad_h // 2
            rows.append(p.as_list(left, up))
        return torch.tensor(rows, dtype=torch.int32)

    def __call__(self, img_u8, lab_u8):
        """img_u8 [B,Hs,Ws,3] uint8, lab_u8 [B,Hs,Ws] uint8 (raw label ids) or None, both already on the device."""
        B, Hs, Ws, _ = img_u8.shape
        params = self.plan([(Ws, Hs)] * B)
        if self.lut is not None and self.lut.device != img_u8.device:
            self.lut = self.lut.to(img_u8.device)
        img, lab = K.augment_batch(img_u8.contiguous(), None if lab_u8 is None else lab_u8.contiguous(), self.lut,
                                   params, (self.target_h, self.target_w), self.div, self.mean, self.std)
        return {'img': img, 'labelmap': lab, 'aug_params': params}
