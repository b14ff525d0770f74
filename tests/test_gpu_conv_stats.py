"""BatchNorm statistics from the convolution epilogue ON THE MI355X against fp64 statistics of the stored output (VERDICT r4 weak 1a:
until round 5 this comparison ran on the CPU emulation only -- tests/test_emu_conv_stats.py -- and the hardware test compared
8-row with 4-row records, i.e. the epilogue with itself). For every `_st` entry point (cseg_conv3x3_split_fwd_st incl. the 8-row
tiles and the head kernel, cseg_conv1x1_split_fwd_st, cseg_conv3x3_s2_split_fwd_st) at the shapes of the benched step (BASELINE
configs[1]: batch 8, branches 48 x 128 x 256 ... 384 x 16 x 32, the 720-channel head) and at ragged ones, the segment records must
finalise (cseg_bn_tiles_finalize, cseg_bn_tiles_moments) to the mean / invstd / running statistics / fp64 moments torch computes in
float64 from the values the kernel stored. Reference semantics: nn.SyncBatchNorm / BatchNorm2d in training mode,
lib/models/tools/module_helper.py:35-39 of the reference. Tolerance: 2e-6 relative (fp32 records, fp64 combination)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NT_SB8 = 0x109
CASES = [
    # kind, B, Cin, Cout, H, W, nt, bias, env
    ("c3", 8, 48, 48, 128, 256, 0, False, {}),                       # finest branch: 8-row tiles (conv3x3_sb16r_kernel), T = 4096
    ("c3", 8, 48, 48, 128, 256, 0, False, {"CSEG_SB16_ROWS8": "0"}),  # the 4-row persistent kernel on the same layer
    ("c3", 8, 96, 96, 64, 128, 0, False, {}),                        # conv3x3_sb_kernel<6>
    ("c3", 8, 192, 192, 32, 64, 3, False, {}),                       # streamed weights, explicit tiling
    ("c3", 8, 384, 384, 16, 32, 3, False, {}),                       # half-empty 64-pixel segments
    ("c3", 8, 64, 64, 128, 256, 0, False, {}),                       # layer 1 (four channel tiles per block)
    ("c3", 2, 720, 720, 128, 256, NT_SB8, True, {}),                 # the head kernel (sb8), with bias
    ("c3", 3, 48, 48, 70, 100, 0, False, {}),                        # ragged: rows % 8 = 6, last segment 36 columns
    ("c3", 3, 48, 48, 70, 100, 0, False, {"CSEG_SB16_ROWS8": "2"}),
    ("c1", 2, 720, 720, 128, 256, 0, True, {}),                      # projection head 1x1
    ("c1", 8, 96, 48, 64, 128, 0, False, {}),                        # exchange unit 1x1
    ("c1", 2, 64, 256, 5, 20, 0, False, {}),                         # 100 flat pixels per image
    ("s2", 8, 48, 96, 64, 128, 0, False, {}),                        # exchange unit stride 2 (output 64 x 128)
    ("s2", 2, 48, 96, 5, 40, 0, False, {}),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%dx%d->%d-%dx%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "-" + ",".join(c[8].values()) if c[8] else ""))
def test_epilogue_statistics_equal_fp64_statistics_of_the_stored_output(case, monkeypatch):
    kind, B, Cin, Cout, H, W, nt, bias, env = case
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from contrastiveseg_amd import kernels as K
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(K, "SPLIT_ARITH", "f16x3")
    monkeypatch.setattr(K, "SPLIT_WEIGHTS", K.SplitWeights())
    monkeypatch.setattr(K, "CONV_EPILOGUE_STATS", True)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 + Cin + H)
    k = 1 if kind == "c1" else 3
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5).to(dev)
    bvec = (torch.randn(Cout, generator=g) * 0.5).to(dev) if bias else None
    # a mean that is NOT small against the spread (per-segment centring is what keeps the fp32 records exact there)
    if kind == "s2":
        x = (torch.randn(B, Cin, 2 * H, 2 * W, generator=g) + 0.7).to(dev)
        y = K.conv3x3_s2_run(x, w, want_stats=True)
    elif kind == "c1":
        x = (torch.randn(B, Cin, H, W, generator=g) + 0.7).to(dev)
        y = K.conv1x1_sb_run(x, w, False, bvec, want_stats=True)
    else:
        x = (torch.randn(B, Cin, H, W, generator=g) + 0.7).to(dev)
        y = K.conv3x3_sb_run(x, w, False, bvec, nt, want_stats=True)
    st = K.known_tile_stats(y)
    assert st is not None and st.shape[0] == Cout and st.shape[2] == 4
    n = y.numel() // Cout
    assert abs(float(st[:, :, 0].double().sum()) - Cout * n) < 0.5, "segment counts do not add up to the tensor"
    # the stored values are what the network uses: a cheap sanity check of them against a strict fp32 convolution (MIOpen)
    ref32 = torch.nn.functional.conv2d(x, w, bvec, 2 if kind == "s2" else 1, 0 if kind == "c1" else 1)
    assert float((y - ref32).abs().max()) <= 1e-4 * float(ref32.abs().max())
    rm0, rv0 = torch.randn(Cout, generator=g).to(dev), (torch.rand(Cout, generator=g) + 0.5).to(dev)
    rm_a, rv_a, nb_a = rm0.clone(), rv0.clone(), torch.tensor(3, device=dev)
    mi_t = K.bn_tiles_finalize(st, 1e-5, 0.1, rm_a, rv_a, nb_a)
    yd = y.double().transpose(0, 1).reshape(Cout, -1)
    mean64, var64 = yd.mean(1), yd.var(1, unbiased=False)
    inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
    e_mean = float((mi_t[:, 0].double() - mean64).abs().max()) / max(1.0, float(mean64.abs().max()))
    e_inv = float((mi_t[:, 1].double() / inv64 - 1).abs().max())
    assert e_mean <= 2e-6 and e_inv <= 2e-6, (e_mean, e_inv)
    # running statistics as nn.BatchNorm2d updates them (momentum 0.1, unbiased variance)
    rm_ref = 0.9 * rm0.double() + 0.1 * mean64
    rv_ref = 0.9 * rv0.double() + 0.1 * var64 * n / (n - 1)
    assert int(nb_a) == 4
    assert float((rm_a.double() - rm_ref).abs().max()) <= 2e-6 * max(1.0, float(rm_ref.abs().max()))
    assert float((rv_a.double() / rv_ref - 1).abs().max()) <= 2e-6
    # the fp64 moments a SyncBN exchange all-reduces: (sum, sum of squares) per channel + the element count
    mo = K.bn_tiles_moments(st)
    assert mo.shape == (Cout + 1, 2) and float(mo[-1, 0]) == n
    s1, s2 = yd.sum(1), (yd * yd).sum(1)
    assert float((mo[:-1, 0] - s1).abs().max()) <= 2e-6 * float(s1.abs().max())
    assert float((mo[:-1, 1] / s2 - 1).abs().max()) <= 2e-6
    print(case[:6], "mean err %.1e invstd err %.1e (relative, vs float64 of the stored output)" % (e_mean, e_inv))
