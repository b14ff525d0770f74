"""
ORACLE -- test infrastructure only. CPU (numpy, float64) restatement of the reference's training data chain for the
accelerated configuration, applied FORWARD the way the reference applies it (the HIP kernel walks it backwards):

  RandomResize   lib/datasets/tools/cv2_aug_transforms.py:327-443  cv2.resize(img, size, INTER_CUBIC).astype(uint8),
                                                                    cv2.resize(label, size, INTER_NEAREST)
  RandomCrop     :504-603   x[up:up+h, left:left+w]
  RandomHFlip    :143-209   cv2.flip(x, 1)
  RandomBrightness :305-325 clip(around(img + shift), 0, 255).astype(uint8)
  ToTensor / Normalize / ToLabel / ReLabel(255,-1)   lib/datasets/tools/transforms.py:15-103
  label encoding  lib/datasets/loader/default_loader.py:94-106 (label_list) -- applied BEFORE the augmentation there;
                  nearest resampling, crop and flip commute with a per-pixel look-up, so it is applied last here too
  collate        lib/datasets/tools/collate.py:37-175 (fix_size, only_pad): F.pad(img, value=0), F.pad(label, value=-1)

Third-party arithmetic: cv2.resize belongs to OpenCV (opencv-python, unpinned in the reference's requirements.txt and NOT
installed in the build image). Its published algorithm (modules/imgproc/src/resize.cpp) is restated here: source
coordinate (dst + 0.5) * scale - 0.5 with scale = 1 / (dsize / ssize), cubic kernel with A = -0.75 on taps
floor(src) - 1 .. + 2, replicated border; INTER_NEAREST = min(floor(dst * scale), ssize - 1). On uint8 OpenCV evaluates
the cubic in 11-bit fixed point with a separable horizontal-then-vertical pass; this restatement (and the kernel) use
floating point with the same pass order, then round half up and saturate. PARITY UNPINNED for the cubic pixel values:
no cv2 to produce vectors with (expected differences: one grey level where the roundings differ). The random decisions
ARE pinned: tests/test_gpu_aug_host.py runs the reference's own transform classes (cv2 stubbed) under a seed and compares
every decision with contrastiveseg_amd/lib/datasets/tools/gpu_aug.py.
"""
import numpy as np


def cubic_weights(t):
    A = -0.75
    w0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    w1 = ((A + 2) * t - (A + 3)) * t * t + 1
    w2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    return np.stack([w0, w1, w2, 1 - w0 - w1 - w2], axis=-1)


def _taps(n_dst, n_src):
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = (np.arange(n_dst) + 0.5) * scale - 0.5
    f = f.astype(np.float32).astype(np.float64)            # the kernel carries the coordinate in fp32
    s = np.floor(f)
    w = cubic_weights((f - s).astype(np.float32).astype(np.float64))
    idx = np.clip(s[:, None].astype(np.int64) - 1 + np.arange(4)[None, :], 0, n_src - 1)
    return idx, w


def resize_cubic_u8(img, size):
    """img [H,W,C] uint8 -> [Hr,Wr,C] uint8, size = (Wr, Hr)."""
    Wr, Hr = size
    H, W = img.shape[:2]
    if (Wr, Hr) == (W, H):
        return img.copy()
    ix, wx = _taps(Wr, W)
    iy, wy = _taps(Hr, H)
    x = img.astype(np.float64)
    rows = (x[:, ix, :] * wx[None, :, :, None]).sum(2)           # horizontal pass: [H, Wr, C]
    out = (rows[iy, :, :] * wy[:, :, None, None]).sum(1)         # vertical pass:   [Hr, Wr, C]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def resize_nearest(x, size):
    Wr, Hr = size
    H, W = x.shape[:2]
    if (Wr, Hr) == (W, H):
        return x.copy()
    sx = np.minimum(np.floor(np.arange(Wr) * (1.0 / (float(Wr) / W))).astype(np.int64), W - 1)
    sy = np.minimum(np.floor(np.arange(Hr) * (1.0 / (float(Hr) / H))).astype(np.int64), H - 1)
    return x[sy][:, sx]


def apply_chain(img, lab, row, target_wh, div, mean, std, lut=None):
    """One sample. row = the 12-int parameter record of include/cseg_hip.h. Returns (img f32 [3,Ht,Wt], lab i64 [Ht,Wt])."""
    Wr, Hr, x_off, y_off, tw, th, flip, shift, left, up = [int(v) for v in row[:10]]
    Wt, Ht = target_wh
    im = resize_cubic_u8(img, (Wr, Hr))
    lb = resize_nearest(lab, (Wr, Hr)) if lab is not None else None
    im = im[y_off:y_off + th, x_off:x_off + tw]
    if lb is not None:
        lb = lb[y_off:y_off + th, x_off:x_off + tw]
    if flip:
        im = im[:, ::-1]
        if lb is not None:
            lb = lb[:, ::-1]
    im = np.clip(np.around(im.astype(np.float32) + shift), 0, 255).astype(np.uint8)
    t = im.astype(np.float32).transpose(2, 0, 1) / np.float32(div)
    t = (t - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    out = np.zeros((3, Ht, Wt), np.float32)
    out[:, up:up + th, left:left + tw] = t
    out_l = None
    if lb is not None:
        lb = lb.astype(np.int64)
        if lut is not None:
            lb = np.asarray(lut, np.int64)[lb]
        lb[lb == 255] = -1
        out_l = np.full((Ht, Wt), -1, np.int64)
        out_l[up:up + th, left:left + tw] = lb
    return out, out_l
