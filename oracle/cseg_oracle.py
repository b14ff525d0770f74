"""
ORACLE -- test infrastructure only.

CPU (numpy, float64) restatement of the contrastive-training hot path of tfzhou/ContrastiveSeg.
Nothing in the product package (`contrastiveseg_amd/`) may import this file; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as the checker.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Pinning: `oracle/make_golden.py` imports the reference itself (CPU, with import shims) in the build
container, runs it on seeded inputs and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this restatement against those vectors. The reference ships no tests or golden vectors of its
own for this path, so the reference-run-here fixtures are the pin.

Third-party arithmetic restated here because it is not under /root/reference:
  * PyTorch (requirements.txt:16, torch>=1.7.0; container has 2.10.0): F.interpolate nearest (legacy) and
    bilinear(align_corners=True), nn.CrossEntropyLoss(weight, ignore_index, mean), torch.max(dim) tie rule,
    torch.randperm / mt19937 (see oracle/mt19937_randperm.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------
# RNG: torch default CPU generator replica (C, see mt19937_randperm.c)
# ----------------------------------------------------------------------------------------------
def _load_c():
    so = os.path.join(_HERE, "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.mt_state_size.restype = ctypes.c_size_t
    lib.mt_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.mt_next.argtypes = [ctypes.c_void_p]
    lib.mt_next.restype = ctypes.c_uint32
    lib.mt_randperm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    return lib


class TorchCpuRng:
    """torch.manual_seed(seed) followed by torch.randperm(n) calls, restated."""

    def __init__(self, seed):
        self._lib = _load_c()
        self._state = ctypes.create_string_buffer(self._lib.mt_state_size())
        self._lib.mt_seed(self._state, ctypes.c_uint32(seed & 0xFFFFFFFF))

    def raw32(self):
        return int(self._lib.mt_next(self._state))

    def randperm(self, n):
        out = np.empty(int(n), dtype=np.int64)
        self._lib.mt_randperm(self._state, ctypes.c_int64(int(n)), out.ctypes.data_as(ctypes.c_void_p))
        return out


# ----------------------------------------------------------------------------------------------
# Resampling primitives (PyTorch semantics)
# ----------------------------------------------------------------------------------------------
def nearest_src_index(out_size, in_size):
    """Legacy 'nearest' source index: src = min(floor(dst * float32(in/out)), in-1).
    Used by lib/loss/loss_contrast.py:131-134 and lib/loss/loss_helper.py:208-212."""
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def nearest_downsample_labels(labels, h, w):
    """lib/loss/loss_contrast.py:131-134: labels -> float -> interpolate(nearest) -> long."""
    labels = np.asarray(labels)
    iy = nearest_src_index(h, labels.shape[-2])
    ix = nearest_src_index(w, labels.shape[-1])
    return labels[..., iy[:, None], ix[None, :]].astype(np.int64)


def bilinear_align_corners(x, out_h, out_w):
    """F.interpolate(mode='bilinear', align_corners=True) on [..., h, w] (float64 math).
    lib/models/nets/hrnet.py:86-91, lib/loss/loss_contrast.py:180."""
    x = np.asarray(x, dtype=np.float64)
    in_h, in_w = x.shape[-2:]

    def axis(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = np.arange(n_out, dtype=np.float64) * scale
        i0 = np.floor(src).astype(np.int64)
        i0 = np.minimum(i0, n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        lam = src - i0
        return i0, i1, lam

    y0, y1, ly = axis(in_h, out_h)
    x0, x1, lx = axis(in_w, out_w)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    rows = top * (1 - ly)[:, None] + bot * ly[:, None]
    return rows[..., :, x0] * (1 - lx) + rows[..., :, x1] * lx


def upcat(feats, h=None, w=None):
    """lib/models/nets/hrnet.py:86-91: bilinear-upsample every map to the first map's size, concat on C."""
    if h is None:
        h, w = feats[0].shape[-2:]
    outs = [np.asarray(feats[0], dtype=np.float64)]
    for f in feats[1:]:
        outs.append(bilinear_align_corners(f, h, w))
    return np.concatenate(outs, axis=1)


def argmax_first(seg):
    """torch.max(seg, 1)[1] (lib/loss/loss_contrast.py:183): first index among ties."""
    return np.argmax(np.asarray(seg), axis=1).astype(np.int64)


# ----------------------------------------------------------------------------------------------
# Cross entropy  (lib/loss/loss_helper.py:169-212  FSCELoss; :301-313 FSAuxCELoss)
# ----------------------------------------------------------------------------------------------
def weighted_ce(logits, target, weight=None, ignore_index=-1, return_grad=False):
    """nn.CrossEntropyLoss(weight, ignore_index, reduction='mean') on logits [B,K,H,W], target [B,H,W]."""
    x = np.asarray(logits, dtype=np.float64)
    t = np.asarray(target)
    B, K = x.shape[:2]
    w = np.ones(K) if weight is None else np.asarray(weight, dtype=np.float64)
    m = x.max(axis=1, keepdims=True)
    lse = np.log(np.exp(x - m).sum(axis=1, keepdims=True)) + m
    logp = x - lse
    valid = t != ignore_index
    assert np.all((t[valid] >= 0) & (t[valid] < K)), "target out of range"
    tc = np.where(valid, t, 0)
    picked = np.take_along_axis(logp, tc[:, None], axis=1)[:, 0]
    wt = w[tc] * valid
    den = wt.sum()
    loss = -(wt * picked).sum() / den
    if not return_grad:
        return loss
    p = np.exp(logp)
    onehot = np.zeros_like(x)
    np.put_along_axis(onehot, tc[:, None], 1.0, axis=1)
    grad = (p - onehot) * (wt / den)[:, None]
    return loss, grad


def upsample_ce(seg, target, weight=None, ignore_index=-1):
    """lib/loss/loss_contrast.py:180-181: pred = bilinear(seg -> target size); FSCELoss(pred, target)."""
    H, W = target.shape[-2:]
    return weighted_ce(bilinear_align_corners(seg, H, W), target, weight, ignore_index)


# ----------------------------------------------------------------------------------------------
# Hard anchor sampling  (lib/loss/loss_contrast.py:30-89)
# ----------------------------------------------------------------------------------------------
class NeverTouched(Exception):
    """lib/loss/loss_contrast.py:75-77 ('this shoud be never touched')."""


def keep_rule(num_hard, num_easy, n_view):
    """lib/loss/loss_contrast.py:66-77."""
    if num_hard >= n_view / 2 and num_easy >= n_view / 2:
        kh = n_view // 2
        ke = n_view - kh
    elif num_hard >= n_view / 2:
        ke = num_easy
        kh = n_view - ke
    elif num_easy >= n_view / 2:
        kh = num_hard
        ke = n_view - kh
    else:
        raise NeverTouched((num_hard, num_easy, n_view))
    return kh, ke


def hard_anchor_sampling(labels, predict, max_samples, max_views, ignore_label, rng):
    """labels (= reference's y_hat, ground truth), predict (= reference's y): int [B, P].
    Returns (segments, n_view) with segments = list of (image, class, pixel_indices[n_view]) in the
    reference's image-major / class-ascending order, or (None, 0) when no class qualifies (:44-45)."""
    labels = np.asarray(labels)
    predict = np.asarray(predict)
    B = labels.shape[0]
    classes = []
    total = 0
    for ii in range(B):
        this_y = labels[ii]
        cs = [c for c in np.unique(this_y) if c != ignore_label]
        cs = [c for c in cs if int((this_y == c).sum()) > max_views]
        classes.append(cs)
        total += len(cs)
    if total == 0:
        return None, 0
    n_view = min(max_samples // total, max_views)
    segments = []
    for ii in range(B):
        for c in classes[ii]:
            hard = np.nonzero((labels[ii] == c) & (predict[ii] != c))[0]
            easy = np.nonzero((labels[ii] == c) & (predict[ii] == c))[0]
            kh, ke = keep_rule(len(hard), len(easy), n_view)
            perm = rng.randperm(len(hard))
            hard = hard[perm[:kh]]
            perm = rng.randperm(len(easy))
            easy = easy[perm[:ke]]
            segments.append((ii, int(c), np.concatenate([hard, easy]).astype(np.int64)))
    return segments, n_view


def gather_anchors(embed, segments, n_view):
    """lib/loss/loss_contrast.py:141-142 + :85-87: X_[ptr] = X[ii, indices, :] from NCHW embeddings.
    Returns X_ [T, n_view, D] (float64) and y_ [T]."""
    embed = np.asarray(embed, dtype=np.float64)
    B, D = embed.shape[:2]
    flat = embed.reshape(B, D, -1)
    T = len(segments)
    X = np.zeros((T, n_view, D))
    y = np.zeros(T)
    for a, (ii, c, idx) in enumerate(segments):
        X[a] = flat[ii][:, idx].T
        y[a] = c
    return X, y


# ----------------------------------------------------------------------------------------------
# Contrastive term  (lib/loss/loss_contrast.py:91-128 ; lib/loss/loss_contrast_mem.py:91-152)
# ----------------------------------------------------------------------------------------------
def _contrast_core(A, ya, C, yc, temperature, base_temperature, return_grad):
    """Rows = anchors A [N,D] with labels ya; columns = contrast set C [M,D] with labels yc.
    Column index i of row i is removed from the positives (the scatter_ self-mask, :111-114 / mem :134-138);
    negatives are all columns with a different label. Denominator = this pair + all negatives (:119-121)."""
    N, M = A.shape[0], C.shape[0]
    S = A @ C.T / temperature
    m = S.max(axis=1, keepdims=True)
    L = S - m
    same = ya[:, None] == yc[None, :]
    neg = ~same
    pos = same.copy()
    idx = np.arange(N)
    assert N <= M, "scatter_(1, arange(N)) needs N <= M"
    pos[idx, idx] = False
    E = np.exp(L)
    Neg = (E * neg).sum(axis=1, keepdims=True)
    logp = L - np.log(E + Neg)
    P = pos.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean_log_prob_pos = (pos * logp).sum(axis=1) / P
    coef = temperature / base_temperature
    loss = (-coef * mean_log_prob_pos).mean()
    if not return_grad:
        return loss
    c = coef / N
    with np.errstate(invalid="ignore", divide="ignore"):
        invP = 1.0 / P
        inv_den = pos / (E + Neg)
        R = inv_den.sum(axis=1, keepdims=True)
        G = c * invP[:, None] * (-(pos * Neg) / (E + Neg) + neg * E * R)
    return loss, G


def contrastive_self(feats_, labels_, temperature, base_temperature, return_grad=False):
    """lib/loss/loss_contrast.py:91-128. feats_ [T, V, D], labels_ [T]. Rows are view-major (:98).
    With return_grad also returns dLoss/dfeats_ [T, V, D]."""
    feats_ = np.asarray(feats_, dtype=np.float64)
    T, V, D = feats_.shape
    F = np.concatenate([feats_[:, v, :] for v in range(V)], axis=0)
    y = np.tile(np.asarray(labels_), V)
    out = _contrast_core(F, y, F, y, temperature, base_temperature, return_grad)
    if not return_grad:
        return out
    loss, G = out
    dF = (G + G.T) @ F / temperature
    dfeats = dF.reshape(V, T, D).transpose(1, 0, 2)
    return loss, dfeats


def sample_negative(queue):
    """lib/loss/loss_contrast_mem.py:91-105: classes 1..K-1 packed from row 0, class 0 skipped, the
    trailing cache_size rows stay zero with label 0."""
    queue = np.asarray(queue, dtype=np.float64)
    K, S, D = queue.shape
    X = np.zeros((K * S, D))
    y = np.zeros(K * S)
    ptr = 0
    for ii in range(1, K):
        X[ptr:ptr + S] = queue[ii, :S]
        y[ptr:ptr + S] = ii
        ptr += S
    return X, y


def contrastive_mem(X_anchor, y_anchor, queue, temperature, base_temperature, return_grad=False):
    """lib/loss/loss_contrast_mem.py:107-152 with queue = cat(segment_queue, pixel_queue, dim=1) (:221)."""
    X_anchor = np.asarray(X_anchor, dtype=np.float64)
    T, V, D = X_anchor.shape
    A = np.concatenate([X_anchor[:, v, :] for v in range(V)], axis=0)
    ya = np.tile(np.asarray(y_anchor), V)
    C, yc = sample_negative(queue)
    out = _contrast_core(A, ya, C, yc, temperature, base_temperature, return_grad)
    if not return_grad:
        return out
    loss, G = out
    dA = G @ C / temperature
    return loss, dA.reshape(V, T, D).transpose(1, 0, 2)


# ----------------------------------------------------------------------------------------------
# Full criteria
# ----------------------------------------------------------------------------------------------
def pixel_contrast_loss(embed, target, predict, cfg, rng, queue=None):
    """PixelContrastLoss.forward (lib/loss/loss_contrast.py:130-147; mem: loss_contrast_mem.py:154-171).
    Returns (loss, segments, n_view)."""
    B, D, h, w = embed.shape
    labels = nearest_downsample_labels(target, h, w).reshape(B, -1)
    predict = np.asarray(predict).reshape(B, -1)
    segments, n_view = hard_anchor_sampling(labels, predict, cfg["max_samples"], cfg["max_views"],
                                            cfg["ignore_label"], rng)
    if segments is None:
        raise RuntimeError("no class qualifies: reference crashes at loss_contrast.py:92")
    X, y = gather_anchors(embed, segments, n_view)
    if queue is None:
        loss = contrastive_self(X, y, cfg["temperature"], cfg["base_temperature"])
    else:
        loss = contrastive_mem(X, y, queue, cfg["temperature"], cfg["base_temperature"])
    return loss, segments, n_view


def contrast_ce_loss(seg, embed, target, cfg, rng, with_embed=True, seg_aux=None, queue=None, mem=False):
    """ContrastCELoss.forward (loss_contrast.py:171-189), ContrastAuxCELoss.forward (:213-234) when
    seg_aux is given, mem ContrastCELoss.forward (loss_contrast_mem.py:198-231) when mem=True."""
    ce = upsample_ce(seg, target, cfg.get("ce_weight"), cfg["ignore_label"])
    if seg_aux is not None:
        aux = upsample_ce(seg_aux, target, cfg.get("ce_weight"), cfg["ignore_label"])
        ce = cfg["seg_loss_weight"] * ce + cfg["aux_loss_weight"] * aux
    if mem and queue is None:
        return ce, None, 0
    predict = argmax_first(seg)
    lc, segments, n_view = pixel_contrast_loss(embed, target, predict, cfg, rng, queue=queue)
    w = cfg["loss_weight"] if with_embed else 0.0
    return ce + w * lc, segments, n_view


# ----------------------------------------------------------------------------------------------
# Memory bank update  (segmentor/trainer_contrastive.py:102-138)
# ----------------------------------------------------------------------------------------------
def _l2n(x, axis, eps=1e-12):
    n = np.sqrt((x * x).sum(axis=axis, keepdims=True))
    return x / np.maximum(n, eps)


def dequeue_and_enqueue(keys, labels, segment_queue, segment_queue_ptr, pixel_queue, pixel_queue_ptr,
                        network_stride, memory_size, pixel_update_freq, rng):
    """In-place update of the four queue arrays, quirks included: labels subsampled with network.stride
    (:108); class ids > 0 only (:114); pixel features indexed with raw perm positions, not idxs[perm]
    (:127-130); pixel pointer advances by 1, not K (:138)."""
    keys = np.asarray(keys, dtype=np.float64)
    B, D = keys.shape[:2]
    labels = np.asarray(labels)[:, ::network_stride, ::network_stride]
    for bs in range(B):
        this_feat = keys[bs].reshape(D, -1)
        this_label = labels[bs].reshape(-1)
        ids = [int(x) for x in np.unique(this_label) if x > 0]
        for lb in ids:
            idxs = np.nonzero(this_label == lb)[0]
            feat = this_feat[:, idxs].mean(axis=1)
            ptr = int(segment_queue_ptr[lb])
            segment_queue[lb, ptr, :] = _l2n(feat, 0)
            segment_queue_ptr[lb] = (segment_queue_ptr[lb] + 1) % memory_size
            num_pixel = idxs.shape[0]
            perm = rng.randperm(num_pixel)
            K = min(num_pixel, pixel_update_freq)
            feat = this_feat[:, perm[:K]].T
            ptr = int(pixel_queue_ptr[lb])
            if ptr + K >= memory_size:
                pixel_queue[lb, -K:, :] = _l2n(feat, 1)
                pixel_queue_ptr[lb] = 0
            else:
                pixel_queue[lb, ptr:ptr + K, :] = _l2n(feat, 1)
                pixel_queue_ptr[lb] = (pixel_queue_ptr[lb] + 1) % memory_size


# ----------------------------------------------------------------------------------------------
# Seeded synthetic inputs shared by the golden generator, the tests, smoke() and bench.py
# ----------------------------------------------------------------------------------------------
def synth_case(seed, B, K, H, W, stride, D, blocky=True, n_rect=14, logit_gain=4.0):
    """Deterministic inputs (numpy legacy RandomState => identical on every machine):
    target int64 [B,H,W] in [-1,K), seg float32 [B,K,h,w], embed float32 [B,D,h,w] unit-norm over D.
    blocky: random rectangles so several classes per image pass the `> max_views` filter and both hard and
    easy sets are populated."""
    rs = np.random.RandomState(seed)
    h, w = H // stride, W // stride
    if blocky:
        target = np.full((B, H, W), -1, dtype=np.int64)
        for b in range(B):
            target[b] = rs.randint(0, K)
            for _ in range(n_rect):
                c = rs.randint(-1, K)
                y0, x0 = rs.randint(0, H), rs.randint(0, W)
                hh, ww = rs.randint(H // 8, H // 2 + 1), rs.randint(W // 8, W // 2 + 1)
                target[b, y0:y0 + hh, x0:x0 + ww] = c
    else:
        target = rs.randint(-1, K, size=(B, H, W)).astype(np.int64)
    lab = nearest_downsample_labels(target, h, w)
    onehot = (lab[:, None] == np.arange(K)[None, :, None, None]).astype(np.float32)
    seg = (onehot * logit_gain + rs.standard_normal((B, K, h, w)) * 2.0).astype(np.float32)
    e = rs.standard_normal((B, D, h, w)).astype(np.float32)
    e = e / np.sqrt((e.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    return target, seg, e.astype(np.float32)
