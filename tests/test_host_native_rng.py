"""The native host helper (contrastiveseg_amd/csrc_host/rng_draws.cpp) must be indistinguishable from calling
torch.randperm draw by draw: same prefixes, same generator state afterwards, same selection plans. CPU only."""
import numpy as np
import pytest
import torch

from contrastiveseg_amd import _host


@pytest.fixture(scope="module")
def native():
    from contrastiveseg_amd.csrc_host import build as hb
    hb.build()
    _host._tried = False
    _host._lib = None
    if _host.lib() is None:
        pytest.skip("libcseg_host.so not available")
    return _host


def test_prefixes_and_generator_state_match_torch(native):
    n_list = [0, 1, 2, 7, 0, 1550, 60, 1, 32768, 5]
    keep = [0, 1, 1, 3, 0, 50, 60, 0, 10, 5]
    torch.manual_seed(304)
    got = native.randperm_prefixes(n_list, keep)
    state_native = torch.get_rng_state()
    torch.manual_seed(304)
    want = [torch.randperm(n).numpy()[:k] for n, k in zip(n_list, keep)]
    state_torch = torch.get_rng_state()
    assert torch.equal(state_native, state_torch)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # the stream continues identically
    assert torch.equal(torch.randperm(100), (torch.set_rng_state(state_torch), torch.randperm(100))[1])


def test_selection_plan_identical_with_and_without_native_helper(native, monkeypatch):
    from contrastiveseg_amd.lib.loss.anchor_sampling import plan_selection
    rs = np.random.RandomState(3)
    counts = np.zeros((8, 19, 2), dtype=np.int64)
    counts[:, :, 0] = rs.randint(0, 1700, size=(8, 19))
    counts[:, :, 1] = rs.randint(0, 200, size=(8, 19))
    counts[2, 5] = 0
    torch.manual_seed(11)
    a = plan_selection(counts, 1024, 100)
    sa = torch.get_rng_state()
    monkeypatch.setattr(_host, "_lib", None)          # force the per-draw torch.randperm path
    torch.manual_seed(11)
    b = plan_selection(counts, 1024, 100)
    sb = torch.get_rng_state()
    assert torch.equal(sa, sb)
    assert (a.T, a.n_view) == (b.T, b.n_view)
    assert np.array_equal(a.row_off, b.row_off) and np.array_equal(a.row_img, b.row_img)
    assert np.array_equal(a.row_lab, b.row_lab)


def test_enqueue_plan_identical_with_and_without_native_helper(native, monkeypatch):
    from contrastiveseg_amd.segmentor.trainer_contrastive import plan_enqueue
    rs = np.random.RandomState(4)
    counts = rs.randint(0, 300, size=(4, 19)).astype(np.int64)
    counts[:, 3] = 0
    ptr0 = rs.randint(0, 50, size=19).astype(np.int64)
    torch.manual_seed(5)
    a = plan_enqueue(counts, ptr0, ptr0[::-1].copy(), 50, 10)
    monkeypatch.setattr(_host, "_lib", None)
    torch.manual_seed(5)
    b = plan_enqueue(counts, ptr0, ptr0[::-1].copy(), 50, 10)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
