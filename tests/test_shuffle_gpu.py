"""cs_permutation (climsim_amd/shuffle.py): the keyed one-kernel permutation that stands in for the reference's shuffle stage
(step2_retrain.py:266-277) on rows resident in HBM.  Integer work: the result must be EXACTLY a permutation, for every size."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd import shuffle
    return shuffle


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 1000, 4097, 172800, 173568, (1 << 20) + 1, 10091520])
def test_it_is_a_permutation_for_every_size(S, n):
    p = S.device_permutation(n, seed=1234)
    assert p.dtype == torch.int64 and p.shape == (n,)
    srt = torch.sort(p).values
    assert bool((srt == torch.arange(n, device="cuda")).all())                  # bijection on 0..n-1, bit-exact


def test_keys_determinism_and_mixing(S):
    n = 172800
    a, b, c = S.device_permutation(n, 7), S.device_permutation(n, 7), S.device_permutation(n, 8)
    assert bool((a == b).all()) and float((a == c).float().mean()) < 1e-3            # same key same permutation; another key another one
    i = torch.arange(n, device="cuda")
    for p in (a, c):
        fixed = int((p == i).sum())
        assert fixed <= 12                                                           # a random permutation has ~1 fixed point
        disp = float((p - i).abs().double().mean()) / n
        assert abs(disp - 1.0 / 3.0) < 0.01                                          # mean |p(i) - i| of a uniform permutation is n/3
        # neighbours do not stay neighbours, and the first batch draws from the whole chunk
        assert float(((p[1:] - p[:-1]).abs() == 1).float().mean()) < 1e-3
        first = p[:8192].double()
        assert abs(float(first.mean()) / n - 0.5) < 0.02 and float(first.min()) < 0.01 * n and float(first.max()) > 0.99 * n
    # chunk seeds of one pass are distinct
    seeds = {S.chunk_seed(3, k) for k in range(1000)}
    assert len(seeds) == 1000


def test_out_buffer_and_argument_checks(S):
    from climsim_amd._lib import EngineError
    buf = torch.empty(100, dtype=torch.int64, device="cuda")
    p = S.device_permutation(60, 5, out=buf)
    assert p.data_ptr() == buf.data_ptr() and sorted(p.cpu().tolist()) == list(range(60))
    with pytest.raises(ValueError):
        S.device_permutation(200, 5, out=buf)
    with pytest.raises(EngineError):
        S.device_permutation(0, 5)
