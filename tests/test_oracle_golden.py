"""Pins the CPU restatement (oracle/cseg_oracle.py) against vectors produced by running the reference
itself (oracle/make_golden.py -> tests/golden/*.npz). CPU only."""
import os
import sys

import numpy as np
import pytest

from oracle import cseg_oracle as O
from oracle.make_golden import LOSS_CASES, ENQ_CASES, case_inputs, enq_inputs, enq_init

SMALL = [n for n, c in LOSS_CASES.items() if c["B"] * c["H"] * c["W"] <= 4 * 128 * 256]
FULL = [n for n in LOSS_CASES if n not in SMALL]


def oracle_cfg(c):
    k = c["contrast"]
    return dict(max_samples=k["max_samples"], max_views=k["max_views"], ignore_label=-1,
                temperature=k["temperature"], base_temperature=0.07, loss_weight=k["loss_weight"],
                ce_weight=c["ce_weight"], seg_loss_weight=1.0, aux_loss_weight=0.4)


def run_oracle(c):
    target, seg, embed, extra = case_inputs(c)
    cfg = oracle_cfg(c)
    rng = O.TorchCpuRng(c["torch_seed"])
    queue = None
    mem = c["loss"].startswith("mem_")
    if mem:
        queue = np.concatenate([extra["segment_queue"], extra["pixel_queue"]], axis=1)
    total, segments, n_view = O.contrast_ce_loss(seg, embed, target, cfg, rng,
                                                 with_embed=c.get("with_embed", True),
                                                 seg_aux=extra.get("seg_aux"), queue=queue, mem=mem)
    return total, segments, n_view, (target, seg, embed, extra, cfg, queue)


def check_indices(g, segments, n_view):
    assert n_view == int(g["n_view"])
    assert len(segments) == len(g["anchor_cls"])
    for a, (ii, cc, idx) in enumerate(segments):
        assert cc == int(g["anchor_cls"][a])
        assert np.all(g["anchor_img"][a] == ii)
        assert np.array_equal(idx, g["anchor_pix"][a]), "anchor indices differ in segment %d" % a


@pytest.mark.parametrize("name", SMALL)
def test_loss_and_indices_small(name, golden_dir):
    c = LOSS_CASES[name]
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % name))
    total, segments, n_view, ctx = run_oracle(c)
    target, seg, embed, extra, cfg, queue = ctx
    h, w = seg.shape[-2:]
    assert np.array_equal(O.nearest_downsample_labels(target, h, w).reshape(len(target), -1), g["labels_ds"])
    assert np.array_equal(O.argmax_first(seg).reshape(len(seg), -1), g["predict"])
    check_indices(g, segments, n_view)
    assert abs(total - float(g["total"])) < 2e-5 * max(1.0, abs(total))
    # contrast term alone + analytic gradient rows
    X, y = O.gather_anchors(embed, segments, n_view)
    if queue is None:
        lc, dX = O.contrastive_self(X, y, cfg["temperature"], cfg["base_temperature"], return_grad=True)
    else:
        lc, dX = O.contrastive_mem(X, y, queue, cfg["temperature"], cfg["base_temperature"], return_grad=True)
    assert abs(lc - float(g["contrast"])) < 2e-5 * max(1.0, abs(lc))
    w_ = cfg["loss_weight"] if c.get("with_embed", True) else 0.0
    rows = (w_ * dX).reshape(-1, X.shape[-1])          # (class-major, view) order == golden order
    assert np.allclose(rows, g["d_embed_rows"], rtol=2e-4, atol=2e-7)
    assert float(g["d_embed_rest_absmax"]) == 0.0
    # CE gradient w.r.t. seg through the bilinear upsample (adjoint checked by finite projection)
    H, W = target.shape[-2:]
    _, gup = O.weighted_ce(O.bilinear_align_corners(seg, H, W), target, cfg["ce_weight"], -1, return_grad=True)
    probe = np.random.RandomState(0).standard_normal(seg.shape)
    lhs = (g["d_seg"].astype(np.float64) * probe).sum()
    scale = 1.0 if "seg_aux" not in extra else cfg["seg_loss_weight"]
    rhs = scale * (gup * O.bilinear_align_corners(probe, H, W)).sum()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(rhs))


@pytest.mark.slow
@pytest.mark.parametrize("name", FULL)
def test_loss_and_indices_full_size(name, golden_dir):
    c = LOSS_CASES[name]
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % name))
    total, segments, n_view, _ = run_oracle(c)
    check_indices(g, segments, n_view)
    assert abs(total - float(g["total"])) < 2e-5 * max(1.0, abs(total))


@pytest.mark.parametrize("name", list(ENQ_CASES))
def test_dequeue_and_enqueue(name, golden_dir):
    c = ENQ_CASES[name]
    g = np.load(os.path.join(golden_dir, "%s.npz" % name))
    sq, pq = enq_init(c)
    sq, pq = sq.astype(np.float64), pq.astype(np.float64)
    sp = np.zeros(c["K"], dtype=np.int64)
    pp = np.zeros(c["K"], dtype=np.int64)
    rng = O.TorchCpuRng(c["torch_seed"])
    for r in range(c["rounds"]):
        target, embed = enq_inputs(c, r)
        O.dequeue_and_enqueue(embed, target, sq, sp, pq, pp, c["network_stride"], c["memory_size"],
                              c["pixel_update_freq"], rng)
        assert np.array_equal(sp, g["segment_ptr_%d" % r])
        assert np.array_equal(pp, g["pixel_ptr_%d" % r])
        assert np.allclose(sq, g["segment_queue_%d" % r], rtol=1e-5, atol=1e-6)
        assert np.allclose(pq, g["pixel_queue_%d" % r], rtol=1e-5, atol=1e-6)


# ---- round 4: OpenCV's 11-bit fixed-point INTER_CUBIC on uint8 (oracle/aug_oracle.py; cv2 itself is not installable here) --------
def test_cubic_fixed_point_known_answers():
    """Known-answer vectors of the published rule (modules/imgproc/src/resize.cpp): the Q11 coefficient tables at exact fractions
    (OpenCV's well-known A = -0.75 values), pixels of a 2x upsampling worked out in integers below, a constant image, and the
    vectorised restatement against the independent scalar one on random images (up- and down-scaling, odd sizes)."""
    from oracle import aug_oracle as A
    q = np.rint(A.cubic_coeffs_f32(np.float32([0.0, 0.25, 0.5, 0.75])) * np.float32(2048)).astype(int)
    assert q.tolist() == [[0, 2048, 0, 0], [-216, 1800, 536, -72], [-192, 1216, 1216, -192], [-72, 536, 1800, -216]]
    assert (q.sum(1) == 2048).all()
    # 4 x 4 -> 8 x 4 (width doubled): fx = (d + 0.5) / 2 - 0.5 = -0.25, 0.25, 0.75, 1.25, ...  => (s, t) = (-1, .75), (0, .25), (0, .75), (1, .25) ...
    row = np.array([0, 10, 20, 30], np.uint8)
    img = np.repeat(row[None, :, None], 4, axis=0)                       # every row the ramp; the vertical pass is the identity (t = 0)
    out = A.resize_cubic_u8(img, (8, 4))[:, :, 0]
    # d = 1: taps S[-1->0], S[0], S[1], S[2] = 0, 0, 10, 20 with (-216, 1800, 536, -72): h = 5360 - 1440 = 3920; v = 3920 * 2048;
    #        (v + 2^21) >> 22 = (8028160 + 2097152) >> 22 = 2
    # d = 2: same taps with (-72, 536, 1800, -216): h = 18000 - 4320 = 13680; (13680 * 2048 + 2^21) >> 22 = 7
    # d = 0: taps S[-2->0], S[-1->0], S[0], S[1] = 0, 0, 0, 10 with (-72, 536, 1800, -216): h = -2160 -> (-4423680 + 2097152) >> 22 = -1 -> 0
    # d = 7: s = 3, t = .25: taps S[2], S[3], S[4->3], S[5->3] = 20, 30, 30, 30 with (-216, 1800, 536, -72): h = -4320 + 67920 = 63600 -> 31
    assert out[0].tolist()[:3] == [0, 2, 7] and out[0, 7] == 31 and (out == out[0]).all()
    assert (3920 * 2048 + (1 << 21)) >> 22 == 2 and (13680 * 2048 + (1 << 21)) >> 22 == 7 and (63600 * 2048 + (1 << 21)) >> 22 == 31
    assert np.unique(A.resize_cubic_u8(np.full((8, 8, 1), 200, np.uint8), (16, 12))).tolist() == [200]
    rs = np.random.RandomState(7)
    img = rs.randint(0, 256, size=(13, 17, 3)).astype(np.uint8)
    for size in ((34, 26), (9, 7), (20, 13), (17, 29), (5, 40)):
        assert np.array_equal(A.resize_cubic_u8(img, size), A.resize_cubic_u8_scalar(img, size)), size
    # overshoot saturates (cubic lobes are negative): a step edge stays inside [0, 255]
    step = np.zeros((4, 16, 1), np.uint8)
    step[:, 8:] = 255
    up = A.resize_cubic_u8(step, (48, 4))
    assert up.min() == 0 and up.max() == 255
