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
installed in the build image, no network). Its published algorithm for 8-bit images (modules/imgproc/src/resize.cpp, 4.x:
`resize` -> ResizeFunc<HResizeCubic<uchar,int,short>, VResizeCubic<uchar,int,short,FixedPtCast<int,uchar,22>,...>>) is
restated here BIT FOR BIT (round 4):
  * per destination index d: fx = (float)((d + 0.5) * scale - 0.5) with scale = 1 / ((double)dsize / ssize) in double,
    s = floor(fx), fx -= s (float); taps s - 1 .. s + 2 with replicated border;
  * interpolateCubic(fx) in FLOAT, A = -0.75f, written exactly as in the source (c3 = 1 - c0 - c1 - c2);
  * INTER_RESIZE_COEF_BITS = 11: coefficient k = saturate_cast<short>(c_k * 2048) = round-half-even (cvRound);
  * horizontal pass in int32: h = sum_k S[x_k] * alpha_k; vertical pass in int32: v = sum_k h_k * beta_k;
  * result = saturate_cast<uchar>((v + (1 << 21)) >> 22) (FixedPtCast<int, uchar, 22>: arithmetic shift, round half up).
  INTER_NEAREST: min(floor(dst * scale), ssize - 1).
What "bit for bit" can and cannot mean without the binary: the above is OpenCV's SCALAR path. Builds with universal
intrinsics run the vertical pass of most of each row through VResizeCubicVec_32s8u, which converts the int32 rows to fp32,
multiplies by beta / 2^22 and rounds half-even -- equal to the fixed-point result except where the exact value lies within
~1e-4 of a rounding boundary (a few pixels per megapixel, one grey level). cv2 itself is not available to produce vectors,
so the known-answer vectors of tests/test_oracle_golden.py::test_cubic_fixed_point_known_answers are hand-derivable cases
(coefficient tables at exact fractions, a 2x upsampling of a ramp worked in integers) plus an independent scalar
restatement (`resize_cubic_u8_scalar`, plain Python integers) of the same published rule. The random decisions ARE pinned
against the reference's own classes (tests/test_gpu_aug_host.py).
"""
import numpy as np


COEF_BITS = 11                      # INTER_RESIZE_COEF_BITS
COEF_SCALE = 1 << COEF_BITS


def cubic_coeffs_f32(t):
    """interpolateCubic (resize.cpp), float arithmetic in the source's operation order. t: float32 array -> [..., 4] float32."""
    f = np.float32
    t = np.asarray(t, dtype=f)
    A = f(-0.75)
    x1 = t + f(1)
    c0 = ((A * x1 - f(5) * A) * x1 + f(8) * A) * x1 - f(4) * A
    c1 = ((A + f(2)) * t - (A + f(3))) * t * t + f(1)
    u = f(1) - t
    c2 = ((A + f(2)) * u - (A + f(3))) * u * u + f(1)
    c3 = f(1) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(f)


def _fixed_taps(n_dst, n_src):
    """-> (tap indices [n_dst, 4] clamped to the source, int32 coefficients [n_dst, 4]) of one axis."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f)
    t = (f - s).astype(np.float32)                                   # float subtraction, like `fx -= sx`
    c = cubic_coeffs_f32(t) * np.float32(COEF_SCALE)                 # float product
    coef = np.clip(np.rint(c), -32768, 32767).astype(np.int32)       # saturate_cast<short>: round half to even
    idx = np.clip(s[:, None].astype(np.int64) - 1 + np.arange(4)[None, :], 0, n_src - 1)
    return idx, coef


def resize_cubic_u8(img, size):
    """cv2.resize(img, size, interpolation=cv2.INTER_CUBIC) for uint8 [H,W,C], OpenCV's fixed-point path. size = (Wr, Hr)."""
    Wr, Hr = size
    H, W = img.shape[:2]
    if (Wr, Hr) == (W, H):
        return img.copy()
    ix, ax = _fixed_taps(Wr, W)
    iy, ay = _fixed_taps(Hr, H)
    x = img.astype(np.int64)
    rows = (x[:, ix, :] * ax[None, :, :, None]).sum(2)               # horizontal pass: [H, Wr, C] (fits int32 like OpenCV's WT)
    v = (rows[iy, :, :] * ay[:, :, None, None]).sum(1)               # vertical pass:   [Hr, Wr, C]
    assert np.abs(rows).max() < 2 ** 31 and np.abs(v).max() < 2 ** 31
    return np.clip((v + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS), 0, 255).astype(np.uint8)


def resize_cubic_u8_scalar(img, size):
    """The same published rule once more, written independently of the vectorised form above: plain Python integers and
    struct-packed float32 arithmetic, pixel by pixel. Slow; used for known-answer vectors only."""
    import struct

    def f32(v):
        return struct.unpack("f", struct.pack("f", v))[0]

    def coeffs(t):
        A = -0.75
        x1 = f32(t + 1.0)
        c0 = f32(f32(f32(f32(f32(f32(A * x1) - f32(5 * A)) * x1) + f32(8 * A)) * x1) - f32(4 * A))
        c1 = f32(f32(f32(f32(f32(f32(A + 2) * t) - f32(A + 3)) * t) * t) + 1.0)
        u = f32(1.0 - t)
        c2 = f32(f32(f32(f32(f32(f32(A + 2) * u) - f32(A + 3)) * u) * u) + 1.0)
        c3 = f32(f32(f32(1.0 - c0) - c1) - c2)
        out = []
        for c in (c0, c1, c2, c3):
            p = f32(c * 2048.0)
            r = int(np.rint(np.float64(p)))                           # cvRound: half to even
            out.append(max(-32768, min(32767, r)))
        return out

    def axis(n_dst, n_src):
        scale = 1.0 / (float(n_dst) / float(n_src))
        taps = []
        for d in range(n_dst):
            fx = f32((d + 0.5) * scale - 0.5)
            s = int(np.floor(fx))
            taps.append((s, coeffs(f32(fx - s))))
        return taps

    Wr, Hr = size
    H, W, C = img.shape
    tx, ty = axis(Wr, W), axis(Hr, H)
    out = np.zeros((Hr, Wr, C), np.uint8)
    for y in range(Hr):
        sy, by = ty[y]
        for x in range(Wr):
            sx, bx = tx[x]
            for c in range(C):
                v = 0
                for j in range(4):
                    yy = min(max(sy - 1 + j, 0), H - 1)
                    h = 0
                    for i in range(4):
                        xx = min(max(sx - 1 + i, 0), W - 1)
                        h += int(img[yy, xx, c]) * bx[i]
                    v += h * by[j]
                out[y, x, c] = min(255, max(0, (v + (1 << 21)) >> 22))
    return out


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
