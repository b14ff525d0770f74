"""Host half of hard-anchor sampling.

The device (cseg_classify_partition) produces per-(image, class) hard/easy counts and the stable partition of
pixel indices; this module turns the counts into the selection the reference makes at
lib/loss/loss_contrast.py:35-87, consuming the CPU default generator with `torch.randperm` in exactly the
reference's order (hard then easy, per class ascending, per image; called even for n = 0) so that the mined
indices are bit-identical under the same `torch.manual_seed`.

Pure host integer logic: runs on CPU tensors, no kernels, shared by the single-GPU and the cross-rank paths."""
import numpy as np

from contrastiveseg_amd import _host


class NeverTouched(Exception):
    """The reference raises a bare Exception here ('this shoud be never touched', loss_contrast.py:75-77)."""


class SelectionPlan(object):
    """Rows are in the contrast (view-major) order: row r = v * T + a (loss_contrast.py:98)."""
    __slots__ = ("T", "n_view", "seg_img", "seg_cls", "row_img", "row_off", "row_lab")

    def __init__(self, T, n_view, seg_img, seg_cls, row_img, row_off, row_lab):
        self.T, self.n_view = T, n_view
        self.seg_img, self.seg_cls = seg_img, seg_cls      # [T]
        self.row_img, self.row_off, self.row_lab = row_img, row_off, row_lab   # [N] int32 numpy

    @property
    def N(self):
        return self.T * self.n_view


def keep_rule(num_hard, num_easy, n_view):
    """loss_contrast.py:66-77."""
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
        raise NeverTouched("this shoud be never touched! {} {} {}".format(num_hard, num_easy, n_view))
    return kh, ke


def plan_selection(counts, max_samples, max_views, draw_images=None):
    """counts: CPU int tensor/array [B, K, 2] (hard, easy) for ALL images of the contrast set, image-major.
    Returns a SelectionPlan, or None when no class qualifies (reference returns (None, None), :44-45).
    row_off[r] is the position inside image row_img[r]'s slice of part_idx.
    draw_images=(lo, hi): draw (and consume the generator) only for images lo <= b < hi; rows of other images get
    row_off = -1. Used by the cross-rank contrast set when every rank samples from its own generator stream."""
    cnt = np.asarray(counts, dtype=np.int64)
    B, K, _ = cnt.shape
    tot = cnt.sum(-1)
    qual = tot > max_views                                   # :39
    T = int(qual.sum())
    if T == 0:
        return None
    n_view = min(max_samples // T, max_views)                # :47-48
    if n_view <= 0:
        raise RuntimeError("anchor sampling: {} qualifying (image, class) segments exceed max_samples={}; the "
                           "reference fails on the empty view axis too".format(T, max_samples))
    flat = cnt.reshape(B, 2 * K)
    off = np.cumsum(flat, axis=1) - flat                     # exclusive offsets inside each image's partition
    # Vectorised (round 6: the Python loop over the ~120 segments of a step took 1.7 ms, and the GPU idles from the D2H copy of the counts
    # until the loss kernels that need this plan are enqueued). Segments in the reference's order: image-major, classes ascending.
    bs, cs = np.nonzero(qual)
    nh, ne = cnt[bs, cs, 0], cnt[bs, cs, 1]
    half = n_view / 2                                        # keep_rule() for all segments at once (loss_contrast.py:66-77)
    both = (nh >= half) & (ne >= half)
    only_h = ~both & (nh >= half)
    only_e = ~both & ~only_h & (ne >= half)
    bad = ~(both | only_h | only_e)
    if bad.any():                                            # raises like the reference, also for foreign segments
        i = int(np.nonzero(bad)[0][0])
        keep_rule(int(nh[i]), int(ne[i]), n_view)
    kh = np.where(both, n_view // 2, np.where(only_h, n_view - ne, nh))
    ke = n_view - kh
    seg_img = bs.astype(np.int32)
    seg_cls = cs.astype(np.int32)
    sel = np.full((T, n_view), -1, dtype=np.int64)
    mine = np.ones(T, dtype=bool) if draw_images is None else (bs >= draw_images[0]) & (bs < draw_images[1])
    idx = np.nonzero(mine)[0]
    if idx.size:
        # torch.randperm(num_hard) then torch.randperm(num_easy) per segment (:79-82), same CPU generator, same order,
        # also for n = 0 -- issued as one native call (contrastiveseg_amd/_host.py). kh + ke = n_view for every segment, so the
        # prefixes, back to back, ARE the rows of `sel` (hard picks first, then easy ones) up to the partition offsets.
        n_list = np.stack([nh[idx], ne[idx]], axis=1).reshape(-1)
        keep = np.stack([kh[idx], ke[idx]], axis=1).reshape(-1)
        picks = _host.randperm_prefixes(n_list, keep, flat=True).reshape(idx.size, n_view)
        off_h = off[bs[idx], 2 * cs[idx]][:, None]
        off_e = off[bs[idx], 2 * cs[idx] + 1][:, None]
        hard_col = np.arange(n_view)[None, :] < kh[idx][:, None]
        sel[idx] = picks + np.where(hard_col, off_h, off_e)
    row_off = np.ascontiguousarray(sel.T).reshape(-1).astype(np.int32)     # view-major
    row_img = np.tile(seg_img, n_view)
    row_lab = np.tile(seg_cls, n_view)
    return SelectionPlan(T, n_view, seg_img, seg_cls, row_img, row_off, row_lab)
