"""Many trials per GPU: K engines on K streams, stepped round-robin, must give each trial the result it gets alone."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import mlp_oracle as O  # noqa: E402


def test_trial_pool_matches_sequential_training():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd.hpo import TrialPool, sample_trial
    x, y = O.synth_columns(8192, seed=1)
    xv, yv = O.synth_columns(2048, seed=2)
    trials = [dict(units=(128, 256), activation="relu", optimizer="Adam", batch_size=768),
              dict(units=(256,), activation="leakyrelu", optimizer="RAdam", batch_size=1536),
              dict(units=(128, 128, 128), activation="elu", optimizer="SGD", batch_size=384)]
    pool = TrialPool(trials, seed=5)
    res = pool.fit(x, y, epochs=3, validation_data=(xv, yv), seed=9)
    pool.close()
    assert len(res) == 3
    # sequential reference: same seeds -> same permutations and initial weights; float atomics in the weight gradients
    # make runs differ in the last bits, so losses are compared to 1e-3 relative
    for k, t in enumerate(trials):
        one = TrialPool([t], seed=5 + k)
        # trial k of the pool draws its permutation from seed + 1000*k + epoch: reproduce by shifting the seed
        r = one.fit(x, y, epochs=3, validation_data=(xv, yv), seed=9 + 1000 * k)[0]
        one.close()
        np.testing.assert_allclose(res[k]["history"]["loss"], r["history"]["loss"], rtol=1e-3)
        np.testing.assert_allclose(res[k]["history"]["val_loss"], r["history"]["val_loss"], rtol=1e-3)
        assert res[k]["history"]["loss"][-1] < res[k]["history"]["loss"][0]
        assert res[k]["objective"] == res[k]["history"]["val_loss"][-1]
    rng = np.random.default_rng(0)
    t = sample_trial(rng)
    assert 2 <= len(t["units"]) <= 12 and all(u % 128 == 0 and 128 <= u <= 1024 for u in t["units"])
