"""Pins oracle/mt19937_randperm.c against torch's own CPU generator (the third-party arithmetic behind
lib/loss/loss_contrast.py:79,81 and segmentor/trainer_contrastive.py:127). CPU only."""
import numpy as np
import pytest
import torch

from oracle.cseg_oracle import TorchCpuRng


@pytest.mark.parametrize("seed", [0, 1, 7, 304, 2 ** 31 + 5])
def test_randperm_stream_matches_torch(seed):
    torch.manual_seed(seed)
    rng = TorchCpuRng(seed)
    for n in [0, 1, 2, 10, 37, 0, 1000, 1, 5, 32768, 3, 70001]:
        want = torch.randperm(n).numpy()
        got = rng.randperm(n)
        assert np.array_equal(want, got), (seed, n)


def test_numpy_randomstate_is_the_same_engine():
    # torch.manual_seed(s) == init_genrand(s) == numpy.random.RandomState(s)
    rs = np.random.RandomState(304)
    rng = TorchCpuRng(304)
    a = rs.randint(0, 2 ** 32, size=2000, dtype=np.uint64)
    b = np.array([rng.raw32() for _ in range(2000)], dtype=np.uint64)
    assert np.array_equal(a, b)
