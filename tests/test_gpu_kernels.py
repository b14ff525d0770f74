"""Parity tests proper: the HIP path (through the C-ABI, via contrastiveseg_amd.kernels) against
 (a) golden vectors produced by the reference itself (tests/golden, oracle/make_golden.py),
 (b) the CPU oracle on seeded inputs, and (c) size-independent properties at BASELINE.json's full sizes.
Bars (BASELINE.json north_star): loss scalars within 1e-3 (asserted much tighter), mined anchor indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import cseg_oracle as O
from oracle.make_golden import GRAD_ROW_STEP, GRAD_SEG_STEP, LOSS_CASES, case_inputs

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-4      # relative; the contract is 1e-3
GRAD_RTOL, GRAD_ATOL = 2e-3, 1e-6


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _configer(c):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    k = dict(proj_dim=c["D"], base_temperature=0.07, use_rmi=False, use_lovasz=False, warmup_iters=0)
    k.update(c["contrast"])
    params = {"ce_ignore_index": -1, "ce_reduction": "elementwise_mean"}
    if c["ce_weight"] is not None:
        params["ce_weight"] = list(c["ce_weight"])
    return Configer(config_dict={
        "data": {"num_classes": c["K"]},
        "network": {"loss_weights": {"aux_loss": 0.4, "seg_loss": 1.0}},
        "contrast": k, "loss": {"loss_type": c["loss"], "params": params}})


def _selection_class_major(crit, V, T):
    sel = crit.contrast_criterion.last_selection["sel_pix"].cpu().numpy()
    return sel.reshape(V, T).T          # [T, V] like the golden files


@pytest.mark.parametrize("name", list(LOSS_CASES))
def test_criterion_matches_reference_golden(name, golden_dir):
    dev = _dev()
    from contrastiveseg_amd.lib.loss.loss_manager import SEG_LOSS_DICT
    c = LOSS_CASES[name]
    g = np.load(os.path.join(golden_dir, "loss_%s.npz" % name))
    target, seg, embed, extra = case_inputs(c)
    crit = SEG_LOSS_DICT[c["loss"]](_configer(c)).to(dev)
    t_seg = torch.from_numpy(seg).to(dev).requires_grad_(True)
    t_embed = torch.from_numpy(embed).to(dev).requires_grad_(True)
    preds = {"seg": t_seg, "embed": t_embed}
    for k, v in extra.items():
        preds[k] = torch.from_numpy(v).to(dev)
        if k == "seg_aux":
            preds[k].requires_grad_(True)
    torch.manual_seed(c["torch_seed"])
    total = crit(preds, torch.from_numpy(target).to(dev), with_embed=c.get("with_embed", True))
    total.backward()
    torch.cuda.synchronize()
    # mined anchors: bit exact
    T, V = len(g["anchor_cls"]), int(g["n_view"])
    plan = crit.contrast_criterion.last_selection["plan"]
    assert (plan.T, plan.n_view) == (T, V)
    P = seg.shape[-2] * seg.shape[-1]
    want = g["anchor_img"].astype(np.int64) * P + g["anchor_pix"]
    got = _selection_class_major(crit, V, T)
    assert np.array_equal(got, want), "mined anchor indices differ from the reference"
    assert np.array_equal(plan.seg_cls, g["anchor_cls"])
    # scalars
    ref = float(g["total"])
    assert abs(float(total.detach()) - ref) <= LOSS_TOL * max(1.0, abs(ref)), (float(total.detach()), ref)
    if not c["grads"]:
        return
    if c["grads"] == "subset":
        # headline shapes (BASELINE configs[1]: 8x19x128x256 logits, 256-d embeddings): a strided slice of d_seg, every
        # GRAD_ROW_STEP-th anchor row of d_embed, and the L1 mass of both gradients
        d_seg = t_seg.grad.cpu().numpy()
        sub = d_seg[:, :, ::GRAD_SEG_STEP, ::GRAD_SEG_STEP]
        ref = g["d_seg_s%d" % GRAD_SEG_STEP]
        assert np.allclose(sub, ref, rtol=GRAD_RTOL, atol=GRAD_ATOL), np.abs(sub - ref).max()
        assert abs(np.abs(d_seg.astype(np.float64)).sum() - float(g["d_seg_abs_sum"])) <= 1e-4 * float(g["d_seg_abs_sum"])
        ge = t_embed.grad.cpu().numpy().reshape(embed.shape[0], embed.shape[1], -1)
        img = g["anchor_img"].reshape(-1)[::GRAD_ROW_STEP]
        pix = g["anchor_pix"].reshape(-1)[::GRAD_ROW_STEP]
        rows = ge[img, :, pix]
        ref = g["d_embed_rows_s%d" % GRAD_ROW_STEP]
        assert np.allclose(rows, ref, rtol=GRAD_RTOL, atol=GRAD_ATOL), np.abs(rows - ref).max()
        assert abs(np.abs(ge.astype(np.float64)).sum() - float(g["d_embed_abs_sum"])) <= 1e-4 * float(g["d_embed_abs_sum"])
        if "seg_aux" in extra:
            sub = preds["seg_aux"].grad.cpu().numpy()[:, :, ::GRAD_SEG_STEP, ::GRAD_SEG_STEP]
            assert np.allclose(sub, g["d_seg_aux_s%d" % GRAD_SEG_STEP], rtol=GRAD_RTOL, atol=GRAD_ATOL)
        return
    # gradients
    d_seg = t_seg.grad.cpu().numpy()
    assert np.allclose(d_seg, g["d_seg"], rtol=GRAD_RTOL, atol=GRAD_ATOL), np.abs(d_seg - g["d_seg"]).max()
    ge = t_embed.grad.cpu().numpy().reshape(embed.shape[0], embed.shape[1], -1)
    img, pix = g["anchor_img"].reshape(-1), g["anchor_pix"].reshape(-1)
    rows = ge[img, :, pix]
    assert np.allclose(rows, g["d_embed_rows"], rtol=GRAD_RTOL, atol=GRAD_ATOL), \
        np.abs(rows - g["d_embed_rows"]).max()
    mask = np.ones(ge.shape, dtype=bool)
    mask[img, :, pix] = False
    assert not ge[mask].any()
    if "seg_aux" in extra:
        d_aux = preds["seg_aux"].grad.cpu().numpy()
        assert np.allclose(d_aux, g["d_seg_aux"], rtol=GRAD_RTOL, atol=GRAD_ATOL)


@pytest.mark.parametrize("name", ["small_self", "mid_self", "uniform_self", "odd_stride8_aux"])
def test_classify_partition_matches_oracle(name):
    dev = _dev()
    from contrastiveseg_amd import kernels as K
    c = LOSS_CASES[name]
    target, seg, embed, _ = case_inputs(c)
    B, Kc, h, w = seg.shape
    cp = K.classify_partition(torch.from_numpy(target).to(dev), -1, seg=torch.from_numpy(seg).to(dev), want_maps=True)
    lab = O.nearest_downsample_labels(target, h, w).reshape(B, -1)
    pred = O.argmax_first(seg).reshape(B, -1)
    assert np.array_equal(cp["lab"].cpu().numpy(), lab)
    assert np.array_equal(cp["pred"].cpu().numpy(), pred)
    counts = cp["counts"].cpu().numpy()
    seg_off = cp["seg_off"].cpu().numpy()
    part = cp["part_idx"].cpu().numpy()
    for b in range(B):
        off = 0
        for cls in range(Kc):
            hard = np.nonzero((lab[b] == cls) & (pred[b] != cls))[0]
            easy = np.nonzero((lab[b] == cls) & (pred[b] == cls))[0]
            assert counts[b, cls, 0] == len(hard) and counts[b, cls, 1] == len(easy)
            assert seg_off[b, cls, 0] == off
            assert np.array_equal(part[b, off:off + len(hard)], hard)
            off += len(hard)
            assert seg_off[b, cls, 1] == off
            assert np.array_equal(part[b, off:off + len(easy)], easy)
            off += len(easy)
    # the predict= entry (reference signature) gives the same partition
    cp2 = K.classify_partition(torch.from_numpy(target).to(dev), -1, predict=torch.from_numpy(pred).to(dev),
                               num_classes=Kc, feat_hw=(h, w))
    assert torch.equal(cp2["counts"], cp["counts"])
    part2 = cp2["part_idx"].cpu().numpy()
    for b in range(B):
        n_valid = int(counts[b].sum())          # entries past the last class are unspecified (dropped pixels)
        assert np.array_equal(part2[b, :n_valid], part[b, :n_valid])


def _rand_unit(rs, n, d):
    x = rs.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("T,V,D,K", [(10, 6, 16, 5), (94, 10, 256, 19), (152, 6, 256, 19), (57, 17, 32, 19),
                                     (8, 128, 256, 4)])
def test_contrast_self_matches_oracle(T, V, D, K):
    dev = _dev()
    from contrastiveseg_amd import kernels as Kk
    rs = np.random.RandomState(T * 1000 + V)
    X = _rand_unit(rs, T * V, D).reshape(T, V, D)
    y = rs.randint(0, K, size=T)
    y[:2] = [0, 0]                                   # at least one class with positives in other segments
    want, dX = O.contrastive_self(X, y, 0.1, 0.07, return_grad=True)
    A = torch.from_numpy(np.ascontiguousarray(X.transpose(1, 0, 2)).reshape(T * V, D)).to(dev).requires_grad_(True)
    lab = torch.from_numpy(np.tile(y, V).astype(np.int32)).to(dev)
    loss = Kk.ContrastOnAnchors.apply(A, lab, "self", 0.1, 0.07, None, None, None, None)
    (loss * 0.37).backward()
    assert abs(float(loss) - want) <= 2e-5 * max(1.0, abs(want)), (float(loss), want)
    got = A.grad.cpu().numpy().reshape(V, T, D).transpose(1, 0, 2) / 0.37
    assert np.allclose(got, dX, rtol=2e-3, atol=1e-7), np.abs(got - dX).max()


@pytest.mark.parametrize("N,Kc,ms,D", [(38, 19, 12, 64), (30, 6, 20, 32), (152, 19, 108, 256)])
def test_contrast_bank_matches_oracle(N, Kc, ms, D):
    dev = _dev()
    from contrastiveseg_amd import kernels as Kk
    rs = np.random.RandomState(N + ms)
    X = _rand_unit(rs, N, D).reshape(N, 1, D)
    y = rs.randint(0, Kc, size=N)
    sq = _rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)
    pq = _rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)
    want, dX = O.contrastive_mem(X, y, np.concatenate([sq, pq], axis=1), 0.07, 0.07, return_grad=True)
    A = torch.from_numpy(X.reshape(N, D)).to(dev).requires_grad_(True)
    loss = Kk.ContrastOnAnchors.apply(A, torch.from_numpy(y.astype(np.int32)).to(dev), "bank", 0.07, 0.07, None, None,
                                      torch.from_numpy(sq).to(dev), torch.from_numpy(pq).to(dev))
    loss.backward()
    assert abs(float(loss) - want) <= 2e-5 * max(1.0, abs(want)), (float(loss), want)
    assert np.allclose(A.grad.cpu().numpy(), dX.reshape(N, D), rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("mode,N,M_or_ms,D,grid", [
    ("self", 60, 0, 16, 0),                 # one strip, one tile, rows 60 .. 63 outside
    ("self", 200, 0, 64, 0),                # 4 strips x 4 tiles, every block one cached tile
    ("self", 200, 0, 64, 4),                # grid capped at 4 blocks: one block per strip walks 4 tiles, 2 from LDS + 2 recomputed
    ("bank", 70, 20, 32, 0),                # M = 6 * 2 * 20 = 240: class-0 skip, zero tail, ragged last tile
    ("bank", 152, 108, 256, 0),             # BASELINE "4096-entry bank": M = 4104, 65 column tiles
    ("bank", 152, 108, 256, 6),             # ... with 3 strips x 2 splits: 33 tiles per block, almost all recomputed
    ("plain", 90, 150, 32, 0),
])
def test_contrast_fused_forward_equals_the_three_launches(mode, N, M_or_ms, D, grid, monkeypatch):
    """Round 5: cseg_contrast_fwd_fused (ONE launch: S tiles on the fp32 MFMA, online row statistics by wavefront reductions, in-launch
    hand-off between the blocks of a row strip, positives from LDS or recomputed) against cseg_contrast_fwd (S to HBM, row pass, mean):
    loss, row_stats, row_loss and the stored S equal to fp32 rounding (online rescaling of the log-sum-exp vs max-then-sum), a row
    without positives is NaN in both, two runs on one scratch buffer are bit-identical (fixed summation orders whichever block arrives
    last; the kernel leaves its counters at zero). Reference: lib/loss/loss_contrast.py:91-128, loss_contrast_mem.py:107-152."""
    dev = _dev()
    from contrastiveseg_amd import kernels as Kk
    if grid:
        monkeypatch.setenv("CSEG_CONTRAST_FUSED_GRID", str(grid))
    rs = np.random.RandomState(N + D)
    A = torch.from_numpy(_rand_unit(rs, N, D)).to(dev)
    Kc = 6 if mode != "self" else 5
    y = rs.randint(0, Kc, size=N).astype(np.int32)
    if mode == "self":
        y[7] = 97                               # a class of its own: no positives -> 0/0 = NaN in row 7, like the reference
    lab = torch.from_numpy(y).to(dev)
    kw = {}
    if mode == "bank":
        Kc, ms = (19, M_or_ms) if M_or_ms == 108 else (6, M_or_ms)
        lab = torch.from_numpy(rs.randint(0, Kc, size=N).astype(np.int32)).to(dev)
        kw = dict(segment_queue=torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev),
                  pixel_queue=torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev))
    elif mode == "plain":
        kw = dict(contrast=torch.from_numpy(_rand_unit(rs, M_or_ms, D)).to(dev),
                  c_lab=torch.from_numpy(rs.randint(0, Kc, size=M_or_ms).astype(np.int32)).to(dev))
    desc = Kk._desc({"self": 0, "plain": 1, "bank": 2}[mode], A, lab, 0.1, 0.07, **kw)
    out = {}
    for name, fused in (("three", "0"), ("fused", "1"), ("fused_again", "1")):
        monkeypatch.setattr(Kk, "CONTRAST_FUSED", fused)
        loss, (S, row_stats, row_loss) = Kk.contrast_forward(desc, dev)
        out[name] = [t.cpu().numpy().copy() for t in (loss, row_stats, row_loss, S)]
    ld = (desc.M + 31) // 32 * 32
    for a, b, what in zip(out["three"], out["fused"], ("loss", "row_stats", "row_loss", "S")):
        if what == "S":
            a, b = a.reshape(-1, ld)[:N, :desc.M], b.reshape(-1, ld)[:N, :desc.M]
            assert np.array_equal(a, b), "the stored similarity tiles are the same MFMA products"
            continue
        assert np.array_equal(np.isnan(a), np.isnan(b)), what
        ok = ~np.isnan(a)
        if not ok.any():
            continue
        scale = np.abs(a[ok]).max()
        assert np.abs(a[ok] - b[ok]).max() <= 2e-6 * max(scale, 1e-30) or np.allclose(a[ok], b[ok], rtol=5e-6, atol=0), (what, np.abs(a[ok] - b[ok]).max(), scale)
    if mode == "self":
        assert np.isnan(out["fused"][2][7]) and np.isnan(out["fused"][0][0])
    for a, b in zip(out["fused"][:3], out["fused_again"][:3]):
        assert np.array_equal(a, b, equal_nan=True), "two launches on one scratch buffer must be bit-identical"
    # S_out = NULL: the N x M array never exists (forward-only uses: validation of the contrastive term, memory-bound bank sizes)
    import ctypes
    from contrastiveseg_amd import _hip
    rs2 = torch.empty(N, 4, dtype=torch.float32, device=dev)
    rl2 = torch.empty(N, dtype=torch.float32, device=dev)
    l2 = torch.empty(1, dtype=torch.float32, device=dev)
    scratch = Kk._fused_ws(desc.N, desc.M, dev)
    _hip.call("cseg_contrast_fwd_fused", ctypes.byref(desc), ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(None),
              ctypes.c_void_p(rs2.data_ptr()), ctypes.c_void_p(rl2.data_ptr()), ctypes.c_void_p(l2.data_ptr()), _hip.stream_ptr())
    for a, b in zip(out["fused"][:3], (l2, rs2, rl2)):
        assert np.array_equal(a, b.cpu().numpy(), equal_nan=True), "without the S store the results are the same bits"


def test_contrast_headline_shape_properties():
    """BASELINE.json sizes: 1024 anchors x 4096-entry bank, D=256. Oracle parity on the scalar + properties:
    invariance to a permutation of bank slots inside a class, and plain mode == bank mode on the packed copy."""
    dev = _dev()
    from contrastiveseg_amd import kernels as Kk
    rs = np.random.RandomState(5)
    N, Kc, ms, D = 1024, 19, 108, 256
    A = torch.from_numpy(_rand_unit(rs, N, D)).to(dev)
    y = torch.from_numpy(rs.randint(0, Kc, size=N).astype(np.int32)).to(dev)
    sq = torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev)
    pq = torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev)
    l_bank = Kk.ContrastOnAnchors.apply(A, y, "bank", 0.1, 0.07, None, None, sq, pq)
    Xp, yp = O.sample_negative(np.concatenate([sq.cpu().numpy(), pq.cpu().numpy()], axis=1))
    l_plain = Kk.ContrastOnAnchors.apply(A, y, "plain", 0.1, 0.07, torch.from_numpy(Xp.astype(np.float32)).to(dev),
                                         torch.from_numpy(yp.astype(np.int32)).to(dev), None, None)
    assert abs(float(l_bank) - float(l_plain)) < 1e-6 * abs(float(l_plain))
    want = O.contrastive_mem(A.cpu().numpy().reshape(N, 1, D), y.cpu().numpy(),
                             np.concatenate([sq.cpu().numpy(), pq.cpu().numpy()], axis=1), 0.1, 0.07)
    assert abs(float(l_bank) - want) <= 2e-5 * abs(want)


def test_upsample_concat_matches_torch():
    dev = _dev()
    import torch.nn.functional as F
    from contrastiveseg_amd import kernels as Kk
    torch.manual_seed(0)
    shapes = [(2, 6, 16, 32), (2, 12, 8, 16), (2, 5, 4, 8), (2, 7, 2, 4)]
    for shp in (shapes, [(1, 3, 13, 21), (1, 4, 7, 11), (1, 2, 3, 5)]):
        xs = [torch.randn(*s, device=dev, requires_grad=True) for s in shp]
        out = Kk.upsample_concat(xs)
        h, w = shp[0][2:]
        ref = torch.cat([xs[0]] + [F.interpolate(x, size=(h, w), mode="bilinear", align_corners=True) for x in xs[1:]], 1)
        assert torch.allclose(out, ref, rtol=1e-5, atol=5e-6), (out - ref).abs().max()   # fp32 rounding only
        rec = Kk.known_amax(out)
        if w % 4 == 0 and Kk.split_arith_id():
            # the max|.| record the kernel leaves for the split-operand convolutions behind it: exactly max|out| (bit patterns of floats >= 0)
            assert rec is not None and float(rec.view(torch.float32).max()) == float(out.detach().abs().max())
        else:
            assert rec is None
        g = torch.randn_like(out)
        got = torch.autograd.grad(out, xs, g)
        want = torch.autograd.grad(ref, xs, g)
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), (a - b).abs().max()
        o = O.upcat([x.detach().cpu().numpy() for x in xs])
        assert np.allclose(out.detach().cpu().numpy(), o, rtol=1e-5, atol=5e-6)


def test_fuse_sum_relu_matches_torch(monkeypatch):
    dev = _dev()
    import torch.nn.functional as F
    from contrastiveseg_amd import kernels as Kk
    monkeypatch.setattr(Kk, "FUSE_SUM_AMAX", True)          # (off by default: the record costs more than the pass it saves; the entry point is tested)
    from oracle import cpu_port
    torch.manual_seed(1)
    for (B, C, h, w), n_same, lows in [((2, 6, 16, 32), 1, [(8, 16), (4, 8), (2, 4)]), ((2, 5, 8, 16), 2, [(4, 8), (2, 4)]),
                                       ((1, 3, 13, 21), 3, [(7, 11)]), ((2, 4, 4, 8), 4, [])]:
        same = [torch.randn(B, C, h, w, device=dev, requires_grad=True) for _ in range(n_same)]
        low = [torch.randn(B, C, lh, lw, device=dev, requires_grad=True) for lh, lw in lows]
        out = Kk.fuse_sum_relu(same, low)
        ref = cpu_port.fuse_sum_relu(same, low)
        assert torch.allclose(out, ref, rtol=1e-5, atol=5e-6), (out - ref).abs().max()
        if Kk.split_arith_id():                       # the max|.| record left for the next unit's split-operand convolutions: exactly max|out|
            rec = Kk.known_amax(out)
            assert rec is not None and float(rec.view(torch.float32).max()) == float(out.detach().abs().max())
        g = torch.randn_like(out)
        got = torch.autograd.grad(out, same + low, g)
        want = torch.autograd.grad(ref, same + low, g)
        for a, b in zip(got, want):
            # the ReLU mask may differ where |pre-activation| ~ 1e-7: compare away from those points
            assert (a - b).abs().max() <= 1e-4 * max(1.0, b.abs().max().item()) or \
                ((a - b).abs() > 1e-4).float().mean().item() < 1e-4, (a - b).abs().max()


@pytest.mark.parametrize("B,Kc,h,w,H,W,weighted", [
    (2, 5, 16, 32, 64, 128, True), (2, 19, 13, 21, 97, 161, False), (1, 171, 17, 9, 65, 33, True),
    (2, 7, 24, 40, 24, 40, True),                 # label resolution: one pixel per cell
    (1, 19, 16, 256, 64, 1024, True),             # one full 256-lane block per coarse row (the benched width)
    (1, 4, 8, 300, 32, 1200, True),               # wider than a block: overlapping column blocks (halo lane)
    (1, 3, 5, 7, 64, 96, False),                  # ~16x upsampling: 16 pixels per cell
    (2, 6, 9, 10, 18, 20, True),                  # ~2x
    (3, 9, 11, 7, 11, 7, False)])                 # label resolution, odd sizes, several bands
def test_upsample_ce_matches_torch_and_oracle(B, Kc, h, w, H, W, weighted):
    dev = _dev()
    import torch.nn.functional as F
    from contrastiveseg_amd import kernels as Kk
    rs = np.random.RandomState(H + Kc)
    seg = torch.from_numpy(rs.standard_normal((B, Kc, h, w)).astype(np.float32) * 3).to(dev).requires_grad_(True)
    target = torch.from_numpy(rs.randint(-1, Kc, size=(B, H, W)).astype(np.int64)).to(dev)
    wt = torch.from_numpy((rs.rand(Kc) + 0.5).astype(np.float32)).to(dev) if weighted else None
    loss = Kk.upsample_ce(seg, target, wt, -1)
    (g_mine,) = torch.autograd.grad(loss * 1.7, seg)
    up = F.interpolate(seg, size=(H, W), mode="bilinear", align_corners=True)
    ref = F.cross_entropy(up, target, weight=wt, ignore_index=-1)
    (g_ref,) = torch.autograd.grad(ref * 1.7, seg)
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert torch.allclose(g_mine, g_ref, rtol=1e-3, atol=1e-7), (g_mine - g_ref).abs().max()
    want = O.upsample_ce(seg.detach().cpu().numpy(), target.cpu().numpy(),
                         None if wt is None else wt.cpu().numpy(), -1)
    assert abs(float(loss) - want) <= 1e-5 * max(1.0, abs(want))


def test_missing_gpu_tensor_is_refused():
    _dev()
    from contrastiveseg_amd import kernels as Kk
    with pytest.raises(RuntimeError):
        Kk.upsample_ce(torch.zeros(1, 3, 4, 4), torch.zeros(1, 8, 8, dtype=torch.long))


@pytest.mark.parametrize("name", ["enq_a", "enq_b"])
def test_trainer_enqueue_on_gpu_matches_reference_golden(name, golden_dir):
    """segmentor/trainer_contrastive.py:102-138 of the reference vs the HIP queue kernels + host pointer logic."""
    dev = _dev()
    from oracle.make_golden import ENQ_CASES, enq_init, enq_inputs
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    c = ENQ_CASES[name]
    g = np.load(os.path.join(golden_dir, "%s.npz" % name))
    me = Trainer.__new__(Trainer)
    me.network_stride, me.memory_size, me.pixel_update_freq = c["network_stride"], c["memory_size"], c["pixel_update_freq"]
    sq, pq = enq_init(c)
    sq, pq = torch.from_numpy(sq).to(dev), torch.from_numpy(pq).to(dev)
    sp = torch.zeros(c["K"], dtype=torch.long, device=dev)
    pp = torch.zeros(c["K"], dtype=torch.long, device=dev)
    torch.manual_seed(c["torch_seed"])
    for r in range(c["rounds"]):
        target, embed = enq_inputs(c, r)
        me._dequeue_and_enqueue(torch.from_numpy(embed).to(dev), torch.from_numpy(target).to(dev), sq, sp, pq, pp)
        assert np.array_equal(sp.cpu().numpy(), g["segment_ptr_%d" % r])
        assert np.array_equal(pp.cpu().numpy(), g["pixel_ptr_%d" % r])
        assert np.allclose(sq.cpu().numpy(), g["segment_queue_%d" % r], rtol=1e-5, atol=1e-6)
        assert np.allclose(pq.cpu().numpy(), g["pixel_queue_%d" % r], rtol=1e-5, atol=1e-6)


def test_side_stream_mining_is_equivalent():
    """The overlap path (mining on a side HIP stream behind a 'seg ready' event) gives the same selection and loss."""
    dev = _dev()
    from contrastiveseg_amd.lib.loss.loss_manager import SEG_LOSS_DICT
    c = LOSS_CASES["mid_self"]
    target, seg, embed, _ = case_inputs(c)
    crit = SEG_LOSS_DICT[c["loss"]](_configer(c)).to(dev)
    t_target = torch.from_numpy(target).to(dev)
    res = []
    for use_event in (False, True, True):
        t_seg = torch.from_numpy(seg).to(dev).requires_grad_(True)
        t_embed = torch.from_numpy(embed).to(dev).requires_grad_(True)
        preds = {"seg": t_seg, "embed": t_embed}
        if use_event:
            preds["seg_ready"] = torch.cuda.Event()
            preds["seg_ready"].record()
            _ = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)   # keep the main stream busy
        torch.manual_seed(c["torch_seed"])
        loss = crit(preds, t_target, with_embed=True)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), crit.contrast_criterion.last_selection["sel_pix"].cpu().numpy(),
                    t_embed.grad.cpu().numpy(), t_seg.grad.cpu().numpy()))
    for r in res[1:]:
        assert r[0] == res[0][0]
        assert np.array_equal(r[1], res[0][1])
        assert np.array_equal(r[2], res[0][2]) and np.array_equal(r[3], res[0][3])


def test_contrast_bank_reference_config_size():
    """configs/cityscapes/H_48_D_4_MEM.json of the reference: memory_size 5000 -> 190 000 bank columns, max_views 1,
    tau 0.07, <= 152 anchors. Bank mode (in place) must equal plain mode on the packed copy and the float64 oracle."""
    dev = _dev()
    from contrastiveseg_amd import kernels as Kk
    rs = np.random.RandomState(9)
    N, Kc, ms, D = 152, 19, 5000, 256
    A = torch.from_numpy(_rand_unit(rs, N, D)).to(dev).requires_grad_(True)
    y = torch.from_numpy(rs.randint(0, Kc, size=N).astype(np.int32)).to(dev)
    sq = torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev)
    pq = torch.from_numpy(_rand_unit(rs, Kc * ms, D).reshape(Kc, ms, D)).to(dev)
    l_bank = Kk.ContrastOnAnchors.apply(A, y, "bank", 0.07, 0.07, None, None, sq, pq)
    (g_bank,) = torch.autograd.grad(l_bank, A)
    queue = np.concatenate([sq.cpu().numpy(), pq.cpu().numpy()], axis=1)
    want, dX = O.contrastive_mem(A.detach().cpu().numpy().reshape(N, 1, D), y.cpu().numpy(), queue, 0.07, 0.07,
                                 return_grad=True)
    assert abs(float(l_bank) - want) <= 2e-5 * abs(want), (float(l_bank), want)
    assert np.allclose(g_bank.cpu().numpy(), dX.reshape(N, D), rtol=2e-3, atol=1e-7)
    Xp, yp = O.sample_negative(queue)
    l_plain = Kk.ContrastOnAnchors.apply(A, y, "plain", 0.07, 0.07, torch.from_numpy(Xp.astype(np.float32)).to(dev),
                                         torch.from_numpy(yp.astype(np.int32)).to(dev), None, None)
    assert abs(float(l_bank) - float(l_plain)) <= 1e-6 * abs(float(l_plain))
