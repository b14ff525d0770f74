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
