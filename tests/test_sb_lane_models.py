"""Lane-level numpy models of the split-bf16 MFMA kernels' INDEX LOGIC (csrc/conv3x3_sb.hip, conv3x3_sb_wrw.hip (both
versions), conv1x1_sb.hip, conv1x1_sb_wrw.hip): weight packing order, K-step / tap pairing for the 16-channel tail, LDS
cell addressing, ring slots, fragment shifts, accumulator -> output mapping, split-K partials. Each model restates the
kernel's index arithmetic per (wave, lane group g, lane n, element j) and contracts A[m][(g,j)] * B[(g,j)][n] exactly as
v_mfma_f32_16x16x32_bf16 does, then compares with a direct convolution in fp64 (LDS images are NaN-poisoned where that
catches stale reads). This is how the kernels were checked before their first hardware run; the arithmetic itself
(bf16 pieces, six products) is covered on the GPU by tests/test_gpu_conv3x3_sb.py."""
import numpy as np
import pytest

# ---------------------------------------------------------------- conv3x3_sb.hip (forward / backward-data)
F_TR, F_TC = 4, 64
F_XROWS, F_XCOLS = F_TR + 2, F_TC + 2
F_CELLS = F_XROWS * F_XCOLS

def f_steps(Cin): return (Cin // 32) * 9 + (5 if Cin & 31 else 0)

def f_pack(w, transpose_flip, NT):
    Cout, Cin = w.shape[:2]
    conv_in, conv_out = (Cout, Cin) if transpose_flip else (Cin, Cout)
    n_full, n_steps = conv_in // 32, f_steps(conv_in)
    wp = np.zeros((conv_out // (NT * 16), n_steps, NT, 64, 8), np.float64)
    wf = w.reshape(Cout, Cin, 9)
    for cot in range(wp.shape[0]):
        for ks in range(n_steps):
            for nt in range(NT):
                for lane in range(64):
                    g, n = lane >> 4, lane & 15
                    oc = (cot * NT + nt) * 16 + n
                    if ks < n_full * 9: tap, ic0 = ks % 9, (ks // 9) * 32 + 8 * g
                    else: tap, ic0 = 2 * (ks - n_full * 9) + (g >> 1), n_full * 32 + 8 * (g & 1)
                    for j in range(8):
                        ic = ic0 + j
                        if tap <= 8:
                            wp[cot, ks, nt, lane, j] = wf[oc, ic, tap] if not transpose_flip else wf[ic, oc, 8 - tap]
    return wp

def f_conv(x, wp, NT, Cout):
    B, Cin, H, W = x.shape
    n_full = Cin // 32
    n_chunks = n_full + (1 if Cin & 31 else 0)
    tiles_x, tiles_y = (W + F_TC - 1) // F_TC, (H + F_TR - 1) // F_TR
    y = np.zeros((B, Cout, H, W))
    for b in range(B):
        for cot in range(Cout // (NT * 16)):
            for ty in range(tiles_y):
                for tx in range(tiles_x):
                    x0, y0 = tx * F_TC, ty * F_TR
                    acc = np.zeros((4, 4, NT, 16, 16))       # row(wave), mt, nt, m(pixel), n(co)
                    ks = 0
                    for c in range(n_chunks):
                        full = c < n_full
                        n_oct = 4 if full else 2
                        As = np.zeros((4, F_CELLS, 8))       # octet, cell, j (all three pieces summed = the fp32 value)
                        for oct in range(n_oct):
                            for rc in range(F_CELLS):
                                r, col = divmod(rc, F_XCOLS)
                                yy, xx = y0 + r - 1, x0 + col - 1
                                if 0 <= yy < H and 0 <= xx < W:
                                    As[oct, rc] = x[b, c * 32 + oct * 8: c * 32 + oct * 8 + 8, yy, xx]
                        for s in range(9 if full else 5):
                            for row in range(4):
                                for mt in range(4):
                                    A = np.zeros((16, 4, 8))        # m, g, j
                                    for g in range(4):
                                        if full:
                                            ky, kx = divmod(s, 3); a_off = g * F_CELLS + ky * F_XCOLS + kx
                                        else:
                                            tap = min(2 * s + (g >> 1), 8); ky, kx = divmod(tap, 3)
                                            a_off = (g & 1) * F_CELLS + ky * F_XCOLS + kx
                                        for n in range(16):
                                            cell = a_off + row * F_XCOLS + n + 16 * mt
                                            A[n, g] = As.reshape(-1, 8)[cell]
                                    for nt in range(NT):
                                        Bm = wp[cot, ks, nt].reshape(4, 16, 8)      # g, n, j
                                        acc[row, mt, nt] += np.einsum('mgj,gnj->mn', A, Bm)
                            ks += 1
                    for row in range(4):
                        yy = y0 + row
                        if yy >= H: continue
                        for mt in range(4):
                            for nt in range(NT):
                                for m in range(16):
                                    xx = x0 + 16 * mt + m
                                    if xx < W:
                                        y[b, (cot * NT + nt) * 16: (cot * NT + nt) * 16 + 16, yy, xx] = acc[row, mt, nt, m]
    return y

def f_ref(x, w):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    y = np.zeros((B, Cout, H, W))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum('bchw,oc->bohw', xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx])
    return y


# ---------------------------------------------------------------- conv3x3_sb_wrw.hip, version 1
W1_CO_B, W1_CI_B, W1_SEG, W1_XP, W1_DP, W1_RPU = 48, 64, 64, 40, 72, 8

def w1_splits(B, Cin, Cout, H, W):
    units = B * (W // W1_SEG) * ((H + W1_RPU - 1) // W1_RPU)
    pairs = ((Cin + W1_CI_B - 1) // W1_CI_B) * (Cout // W1_CO_B)
    n = (768 + pairs - 1) // pairs
    return max(1, min(n, 256, units))

def w1_model(x, dy, n_split=None):
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if n_split is None: n_split = w1_splits(B, Cin, Cout, H, W)
    n_cib = (Cin + W1_CI_B - 1) // W1_CI_B
    segs, runs = W // W1_SEG, (H + W1_RPU - 1) // W1_RPU
    n_units = B * segs * runs
    partial = np.zeros((n_split, 9, Cout, Cin))
    for cob in range(Cout // W1_CO_B):
        for cib in range(n_cib):
            for split in range(n_split):
                acc = np.zeros((8, 9, 3, 16, 16))            # wave, tap, cot, m(co), n(ci)
                xs = np.zeros((2, W1_CI_B, 3, W1_XP)); ds = np.zeros((W1_CO_B, W1_DP))
                def x_store(b, x0, row):
                    slot = (row + 3) % 3
                    for s in range(2):
                        for ci in range(W1_CI_B):
                            for i in range(40):
                                px = x0 + 32 * s - 1 + i
                                ok = cib * W1_CI_B + ci < Cin and 0 <= row < H and i < 34 and 0 <= px < W
                                xs[s, ci, slot, i] = x[b, cib * W1_CI_B + ci, row, px] if ok else 0.0
                def d_store(b, x0, row):
                    for co in range(W1_CO_B):
                        ds[co, :64] = dy[b, cob * W1_CO_B + co, row, x0:x0 + 64] if row < H else 0.0
                for unit in range(split, n_units, n_split):
                    t = unit
                    run = t % runs; t //= runs
                    seg = t % segs; b = t // segs
                    x0, ya = seg * W1_SEG, run * W1_RPU
                    yb = min(ya + W1_RPU, H)
                    for r in (ya - 1, ya, ya + 1): x_store(b, x0, r)
                    d_store(b, x0, ya)
                    for row in range(ya, yb):
                        more = row + 1 < yb
                        for wave in range(8):
                            sl, ksl = wave & 3, wave >> 2
                            if cib * W1_CI_B + sl * 16 >= Cin: continue
                            A = np.zeros((3, 16, 4, 8))      # cot, m, g, j
                            for c in range(3):
                                for n in range(16):
                                    for g in range(4):
                                        A[c, n, g] = ds[c * 16 + n, 32 * ksl + 8 * g: 32 * ksl + 8 * g + 8]
                            for ky in range(3):
                                slot = (row + ky - 1 + 3) % 3
                                for kx in range(3):
                                    Bm = np.zeros((4, 16, 8))    # g, n, j
                                    for n in range(16):
                                        for g in range(4):
                                            ent = xs[ksl, sl * 16 + n, slot, 8 * g: 8 * g + 10]     # cell + next dword
                                            Bm[g, n] = ent[kx: kx + 8]
                                    for c in range(3):
                                        acc[wave, ky * 3 + kx, c] += np.einsum('mgj,gnj->mn', A[c], Bm)
                        if more:
                            x_store(b, x0, row + 2); d_store(b, x0, row + 1)
                for sl in range(4):
                    if cib * W1_CI_B + sl * 16 >= Cin: continue
                    tot = acc[sl] + acc[4 + sl]
                    for tp in range(9):
                        for c in range(3):
                            co0 = cob * W1_CO_B + c * 16
                            partial[split, tp, co0:co0 + 16, cib * W1_CI_B + sl * 16: cib * W1_CI_B + sl * 16 + 16] = tot[tp, c]
    dw = partial.sum(0)                                    # [tap][co][ci]
    return dw.transpose(1, 2, 0).reshape(Cout, Cin, 3, 3)

def w_ref(x, dy):
    B, Cin, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    dw = np.zeros((dy.shape[1], Cin, 3, 3))
    for ky in range(3):
        for kx in range(3):
            dw[:, :, ky, kx] = np.einsum('bohw,bchw->oc', dy, xp[:, :, ky:ky + H, kx:kx + W])
    return dw


# ---------------------------------------------------------------- conv3x3_sb_wrw.hip, version 2 (producer / consumer)
W2_CO_B, W2_CI_B, W2_SEG, W2_DP, W2_RPU, W2_XCH = 48, 64, 64, 72, 16, 296

def w2_model(x, dy, n_split):
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    n_cib = (Cin + W2_CI_B - 1) // W2_CI_B
    segs, runs = W // W2_SEG, (H + W2_RPU - 1) // W2_RPU
    n_units = B * segs * runs
    partial = np.zeros((n_split, 9, Cout, Cin))
    for cob in range(Cout // W2_CO_B):
        for cib in range(n_cib):
            for split in range(n_split):
                acc = np.zeros((4, 9, 3, 16, 16))
                xs = np.full((W2_CI_B, 4, 72), np.nan); ds = np.full((2, W2_CO_B, W2_DP), np.nan)
                def x_put(b, x0, row, slot):
                    for ci in range(W2_CI_B):
                        for c in range(18):
                            px = x0 - 4 + 4 * c
                            ok = cib * W2_CI_B + ci < Cin and 0 <= row < H and 0 <= px < W
                            xs[ci, slot, 4 * c: 4 * c + 4] = x[b, cib * W2_CI_B + ci, row, px: px + 4] if ok else 0.0
                def d_put(b, x0, row, buf):
                    for co in range(W2_CO_B):
                        ds[buf, co, :64] = dy[b, cob * W2_CO_B + co, min(row, H - 1), x0: x0 + 64]
                for unit in range(split, n_units, n_split):
                    t = unit
                    run = t % runs; t //= runs
                    seg = t % segs; b = t // segs
                    x0, ya = seg * W2_SEG, run * W2_RPU
                    yb = min(ya + W2_RPU, H)
                    x_put(b, x0, ya - 1, 0); x_put(b, x0, ya, 1); x_put(b, x0, ya + 1, 2); d_put(b, x0, ya, 0)
                    for row in range(ya, yb):
                        k = row - ya
                        # consumers first (they read what was staged before this tick's barrier), then this tick's loader writes
                        for wave in range(4):
                            if cib * W2_CI_B + wave * 16 >= Cin: continue
                            s0, buf = k & 3, k & 1
                            for ks in range(2):
                                A = np.zeros((3, 16, 4, 8))
                                for c in range(3):
                                    for n in range(16):
                                        for g in range(4):
                                            A[c, n, g] = ds[buf, c * 16 + n, 32 * ks + 8 * g: 32 * ks + 8 * g + 8]
                                for ky in range(3):
                                    slot = (s0 + ky) & 3
                                    for kx in range(3):
                                        Bm = np.zeros((4, 16, 8))
                                        for n in range(16):
                                            for g in range(4):
                                                e = 32 * ks + 8 * g
                                                cells = xs[wave * 16 + n, slot, e: e + 16]
                                                Bm[g, n] = cells[kx + 3: kx + 11]
                                        for c in range(3):
                                            acc[wave, ky * 3 + kx, c] += np.einsum('mgj,gnj->mn', A[c], Bm)
                        if row + 1 < yb:
                            x_put(b, x0, row + 2, (k + 3) & 3); d_put(b, x0, row + 1, (k + 1) & 1)
                for wave in range(4):
                    if cib * W2_CI_B + wave * 16 >= Cin: continue
                    for tp in range(9):
                        for c in range(3):
                            co0 = cob * W2_CO_B + c * 16
                            partial[split, tp, co0:co0 + 16, cib * W2_CI_B + wave * 16: cib * W2_CI_B + wave * 16 + 16] = acc[wave, tp, c]
    assert not np.isnan(partial).any()
    return partial.sum(0).transpose(1, 2, 0).reshape(Cout, Cin, 3, 3)

def w2_ref(x, dy):
    B, Cin, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    dw = np.zeros((dy.shape[1], Cin, 3, 3))
    for ky in range(3):
        for kx in range(3):
            dw[:, :, ky, kx] = np.einsum('bohw,bchw->oc', dy, xp[:, :, ky:ky + H, kx:kx + W])
    return dw


# ---------------------------------------------------------------- conv1x1_sb.hip
O_MT_PX=256
def o_steps(c): return (c+31)//32
def o_pack(w, transpose, NT):
    Cout, Cin = w.shape
    conv_in, conv_out = (Cout, Cin) if transpose else (Cin, Cout)
    ns = o_steps(conv_in)
    wp = np.zeros((conv_out//(NT*16), ns, NT, 64, 8))
    for cot in range(wp.shape[0]):
        for ks in range(ns):
            for nt in range(NT):
                for lane in range(64):
                    g, n = lane>>4, lane&15
                    oc = (cot*NT+nt)*16+n
                    for j in range(8):
                        ic = ks*32+8*g+j
                        if ic < conv_in:
                            wp[cot,ks,nt,lane,j] = w[ic,oc] if transpose else w[oc,ic]
    return wp
def o_model(x, wp, NT, Cout):
    B, Cin, P = x.shape
    y = np.zeros((B, Cout, P))
    ns = o_steps(Cin)
    for b in range(B):
        for cot in range(Cout//(NT*16)):
            for tp in range((P+O_MT_PX-1)//O_MT_PX):
                px0 = tp*O_MT_PX
                acc = np.zeros((4,4,NT,16,16))
                for ks in range(ns):
                    As = np.zeros((4, O_MT_PX, 8))
                    for oct in range(4):
                        for p in range(O_MT_PX):
                            for j in range(8):
                                c = ks*32+oct*8+j
                                if px0+p < P and c < Cin: As[oct,p,j] = x[b,c,px0+p]
                    for q in range(4):
                        for mt in range(4):
                            A = np.zeros((16,4,8))
                            for g in range(4):
                                for n in range(16):
                                    A[n,g] = As[g, q*64+16*mt+n]
                            for nt in range(NT):
                                Bm = wp[cot,ks,nt].reshape(4,16,8)
                                acc[q,mt,nt] += np.einsum('mgj,gnj->mn', A, Bm)
                for q in range(4):
                    for mt in range(4):
                        for nt in range(NT):
                            for m in range(16):
                                px = px0+q*64+16*mt+m
                                if px < P: y[b,(cot*NT+nt)*16:(cot*NT+nt)*16+16,px] = acc[q,mt,nt,m]
    return y


# ---------------------------------------------------------------- conv1x1_sb_wrw.hip
Q_CO_T, Q_CI_T, Q_STG, Q_PITCH = 144, 128, 32, 40
def q_model(x, dy, n_split):
    B, Cin, P = x.shape; Cout = dy.shape[1]
    n_cib, n_cob = (Cin+Q_CI_T-1)//Q_CI_T, (Cout+Q_CO_T-1)//Q_CO_T
    spi = P//Q_STG; n_units = B*spi
    partial = np.full((n_split, Cout, Cin), np.nan)
    for cob in range(n_cob):
        for cib in range(n_cib):
            for split in range(n_split):
                u_lo, u_hi = n_units*split//n_split, n_units*(split+1)//n_split
                acc = np.zeros((4,5,4,16,16))
                ds = np.full((2,Q_CO_T,Q_PITCH), np.nan); xs = np.full((2,Q_CI_T,Q_PITCH), np.nan)
                def put(unit, buf):
                    b, p0 = unit//spi, (unit%spi)*Q_STG
                    for r in range(Q_CO_T):
                        ds[buf,r,:32] = dy[b,cob*Q_CO_T+r,p0:p0+32] if cob*Q_CO_T+r < Cout else 0
                    for r in range(Q_CI_T):
                        xs[buf,r,:32] = x[b,cib*Q_CI_T+r,p0:p0+32] if cib*Q_CI_T+r < Cin else 0
                if u_lo < u_hi: put(u_lo, 0)
                for unit in range(u_lo, u_hi):
                    k = unit-u_lo; buf = k&1
                    for wave in range(4):
                        mh, nh = wave&1, wave>>1
                        cot0, cit0 = (5 if mh else 0), nh*4
                        for a in range(5):
                            if a==4 and mh: break
                            A = np.zeros((16,4,8))
                            for n in range(16):
                                for g in range(4): A[n,g] = ds[buf,(cot0+a)*16+n, 8*g:8*g+8]
                            for c in range(4):
                                Bm = np.zeros((4,16,8))
                                for n in range(16):
                                    for g in range(4): Bm[g,n] = xs[buf,(cit0+c)*16+n, 8*g:8*g+8]
                                acc[wave,a,c] += np.einsum('mgj,gnj->mn', A, Bm)
                    if unit+1 < u_hi: put(unit+1, (k+1)&1)
                for wave in range(4):
                    mh, nh = wave&1, wave>>1
                    cot0, cit0 = (5 if mh else 0), nh*4
                    for a in range(5):
                        if a==4 and mh: break
                        for c in range(4):
                            for m in range(16):
                                for n in range(16):
                                    co, ci = cob*Q_CO_T+(cot0+a)*16+m, cib*Q_CI_T+(cit0+c)*16+n
                                    if co < Cout and ci < Cin: partial[split,co,ci] = acc[wave,a,c,m,n]
    assert not np.isnan(partial).any()
    return partial.sum(0)


# ---------------------------------------------------------------- tests
@pytest.mark.parametrize("case", [(48, 48, 5, 8, 3), (32, 96, 3, 68, 6), (16, 48, 6, 4, 3), (144, 144, 4, 10, 9)])
def test_forward_and_backward_data_index_logic(case):
    Cin, Cout, H, W, NT = case
    rs = np.random.RandomState(0)
    x, w = rs.standard_normal((1, Cin, H, W)), rs.standard_normal((Cout, Cin, 3, 3))
    assert np.abs(f_conv(x, f_pack(w, False, NT), NT, Cout) - f_ref(x, w)).max() < 1e-10
    if Cin % 48 == 0:
        NTb = 9 if Cin % 144 == 0 else 6 if Cin % 96 == 0 else 3
        dy = rs.standard_normal((1, Cout, H, W))
        wt = np.flip(w, (2, 3)).transpose(1, 0, 2, 3)
        assert np.abs(f_conv(dy, f_pack(w, True, NTb), NTb, Cin) - f_ref(dy, wt)).max() < 1e-10


@pytest.mark.parametrize("case", [(1, 48, 48, 5, 64, None), (2, 16, 48, 9, 128, 3), (1, 80, 96, 3, 64, None)])
def test_weight_gradient_v1_index_logic(case):
    B, Cin, Cout, H, W, ns = case
    rs = np.random.RandomState(1)
    x, dy = rs.standard_normal((B, Cin, H, W)), rs.standard_normal((B, Cout, H, W))
    assert np.abs(w1_model(x, dy, ns) - w_ref(x, dy)).max() < 1e-10


@pytest.mark.parametrize("case", [(1, 48, 48, 5, 64, 1), (2, 16, 48, 18, 128, 3), (1, 80, 96, 3, 64, 1)])
def test_weight_gradient_v2_index_logic(case):
    B, Cin, Cout, H, W, ns = case
    rs = np.random.RandomState(2)
    x, dy = rs.standard_normal((B, Cin, H, W)), rs.standard_normal((B, Cout, H, W))
    assert np.abs(w2_model(x, dy, ns) - w2_ref(x, dy)).max() < 1e-10


@pytest.mark.parametrize("case", [(48, 64, 300, 4), (144, 48, 256, 3), (16, 128, 40, 8)])
def test_pointwise_index_logic(case):
    Cin, Cout, P, NT = case
    rs = np.random.RandomState(3)
    x, w = rs.standard_normal((1, Cin, P)), rs.standard_normal((Cout, Cin))
    assert np.abs(o_model(x, o_pack(w, False, NT), NT, Cout) - np.einsum("oc,bcp->bop", w, x)).max() < 1e-10
    if Cin % 48 == 0:
        dy = rs.standard_normal((1, Cout, P))
        assert np.abs(o_model(dy, o_pack(w, True, 3), 3, Cin) - np.einsum("oc,bop->bcp", w, dy)).max() < 1e-10


@pytest.mark.parametrize("case", [(2, 48, 64, 64, 3), (1, 144, 160, 96, 2), (1, 16, 304, 32, 1)])
def test_pointwise_weight_gradient_index_logic(case):
    B, Cin, Cout, P, ns = case
    rs = np.random.RandomState(4)
    x, dy = rs.standard_normal((B, Cin, P)), rs.standard_normal((B, Cout, P))
    assert np.abs(q_model(x, dy, ns) - np.einsum("bop,bcp->oc", dy, x)).max() < 1e-10


# ---------------------------------------------------------------- host-side dispatch heuristics (contrastiveseg_amd/kernels.py)
def test_host_tiling_heuristics():
    import torch
    from contrastiveseg_amd import kernels as K
    meta = lambda *s: torch.empty(*s, device="meta")
    # channel tiles per block: largest tiling that still gives >= 256 blocks
    assert K.conv3x3_sb_pick_nt(meta(8, 192, 32, 64), 192) == 3          # 128 blocks at nt 6 (measured 113 us) vs 256 at nt 3 (81 us)
    assert K.conv3x3_sb_pick_nt(meta(8, 96, 64, 128), 96) == 6           # 256 blocks at nt 6 (70 us; nt 3: 89 us)
    assert K.conv3x3_sb_pick_nt(meta(8, 720, 128, 256), 720) == 9
    assert K.conv3x3_sb_pick_nt(meta(8, 48, 128, 256), 48) == 3
    assert K.conv3x3_sb_pick_nt(meta(1, 384, 16, 32), 384) == 3          # never fills the chip: smallest tiling
    # grid sizes used for the "does it fill the chip" gate
    assert K.conv3x3_sb_tiles(meta(8, 720, 128, 256), 720) == 8 * 5 * 32 * 4
    assert K.conv3x3_sb_tiles(meta(2, 48, 16, 32), 48) == 2 * 1 * 4 * 1
    assert K.CONV3X3_SB_MIN_TILES == K.CONV1X1_SB_MIN_TILES == 1          # round 5: every covered shape takes the split kernels (host-bound below ~4 images)
    assert K.conv1x1_sb_tiles(meta(8, 720, 128, 256), 256) == 8 * 2 * 128
    assert K.conv1x1_sb_tiles(meta(8, 720, 128, 256), 720) == 8 * 5 * 128
    # defaults (round 3): every split kernel that won its hardware timing in the round-2 driver pass is on
    assert isinstance(K.CONV3X3_SPLIT_BF16, bool) and K.CONV3X3_SB_BRANCH_CHANNELS[:3] == (48, 64, 96)
    import os
    if not any(k.startswith("CSEG_CONV") for k in os.environ):
        assert K.CONV3X3_SB_WRW and K.CONV3X3_SB_WRW_CHANNELS == (48, 64, 96, 128, 192, 384, 720)      # 64 / 128: round 6 (partly filled channel block)
        assert K.CONV1X1_SPLIT_BF16 and K.CONV1X1_SB_WRW and K.CONV1X1_SB_WRW_MIN_CH == 16     # 256 until the lean loader (round 3)
        assert K.CONV3X3_S2_SPLIT and K.CONV3X3_SB8 and K.CONV3X3_SB_WRW_PAIRS == ((256, 48),)
        if "CSEG_SPARSE_EMBED_GRAD" not in os.environ:
            assert K.SPARSE_EMBED_GRAD


def test_round_aware_split_counts_and_head_kernel_rule():
    """The launch-shape decisions of round 3 that were fitted to hardware timings: pixel splits of the weight
    gradients from the rounds-of-blocks cost models (host functions of libcseg_hip.so, no GPU needed), the 8-row head kernel only for
    launches of at least four rounds."""
    import ctypes
    import os
    import torch
    from contrastiveseg_amd import _hip, kernels as K
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip("libcseg_hip.so not built")
    lib = ctypes.CDLL(_hip.LIB_PATH)
    lib.cseg_conv3x3_sb_wrw_ws_floats.restype = ctypes.c_size_t
    lib.cseg_conv1x1_sb_wrw_ws_floats.restype = ctypes.c_size_t
    lib.cseg_conv3x3_s2_wrw_ws_floats.restype = ctypes.c_size_t
    if "CSEG_SB_WRW_SPLITS" not in os.environ and "CSEG_CONV3X3_SB_WRW_V" not in os.environ:
        splits3 = lambda B, C, H, W: lib.cseg_conv3x3_sb_wrw_ws_floats(B, C, C, H, W) // (9 * C * C)
        assert splits3(8, 720, 128, 256) == 4        # 30-block groups over 8 XCDs: 3 rounds x 64 units (6.86 ms; 5 splits: 7.70, 3: 9.74)
        assert splits3(8, 48, 128, 256) == 256       # one unit per block (50 us; 128 splits: 70)
        assert splits3(8, 96, 64, 128) == 64
        assert splits3(8, 192, 32, 64) == 16
        assert splits3(8, 384, 16, 32) == 4
    assert lib.cseg_conv1x1_sb_wrw_ws_floats(8, 720, 720, 128 * 256) // (720 * 720) == 17       # 510 blocks = 2 rounds (26: a 4th round for 12 blocks)
    assert lib.cseg_conv3x3_s2_wrw_ws_floats(8, 48, 96, 64, 128) // (9 * 48 * 96) == 128        # 256 blocks of one 16-row unit
    assert lib.cseg_conv3x3_s2_wrw_ws_floats(8, 48, 96, 64, 100) == 0                           # output width % 32
    meta = lambda *s: torch.empty(*s, device="meta")
    if K.CONV3X3_SB8 and K.SPLIT_ARITH == "f16x3":
        assert K.conv3x3_sb_head_nt(720, meta(8, 720, 128, 256)) == K.NT_SB8          # 2 560 blocks: 5.28 vs 5.41 ms
        assert K.conv3x3_sb_head_nt(720, meta(1, 720, 128, 256)) == 0                 # 320 blocks: 0.99 vs 0.88 ms
        assert K.conv3x3_sb_head_nt(144, meta(8, 144, 64, 128)) == 0                  # 128 blocks
    assert K.conv3x3_sb_head_nt(96, meta(8, 96, 64, 128)) == 0
    assert K.conv3x3_s2_pick_nt(8, 64, 128, 96) == 6 and K.conv3x3_s2_pick_nt(8, 32, 64, 192) == 3 and K.conv3x3_s2_pick_nt(8, 64, 128, 256) == 4
