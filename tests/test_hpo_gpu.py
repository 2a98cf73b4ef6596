"""Many trials per GPU (climsim_amd/hpo.py: TrialPool): every trial of a pool must get the result it gets alone, and - round 3 -
the result the ORACLE gets (oracle/mlp_oracle.py: bf16-emulating forward / backward, Keras Adam, tfa cyclical schedule) on the
same permutations, including the grouped validation pass (cs_mlp_group_forward)."""
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


def test_trial_pool_follows_the_oracle_including_the_grouped_validation_pass():
    """Two trials of one kernel family (so they step and validate as ONE group) against the oracle: same initial weights
    (glorot seed + trial index), same permutations (seed + 1000 k + epoch, torch.randperm on the device), the reference's
    cyclical schedule, per-epoch training loss (mean of the step losses) and validation loss.  Tolerance: loss curves 2e-2
    relative (accumulation order and bf16 rounding points over 2 x 6 steps; the first epoch's validation loss 1e-2)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd.hpo import TrialPool
    from climsim_amd.mlp import CyclicalLearningRate, glorot_uniform_weights
    x, y = O.synth_columns(4608, seed=11)
    xv, yv = O.synth_columns(1500, seed=12)                 # not a multiple of anything: ragged last validation batch
    trials = [dict(units=(256, 128), activation="relu", optimizer="Adam", batch_size=768),
              dict(units=(128, 256, 128), activation="leakyrelu", optimizer="Adam", batch_size=1152)]
    pool = TrialPool(trials, seed=21)
    assert [len(b) for b in pool.buckets] == [2] and pool.groups[0] is not None
    res = pool.fit(x, y, epochs=2, validation_data=(xv, yv), seed=4)
    pool.close()
    gen = torch.Generator(device="cuda")
    for k, t in enumerate(trials):
        cfg = O.MLPConfig(hidden=tuple(t["units"]), act=t["activation"])
        ws = glorot_uniform_weights(124, t["units"], 120, 8, 21 + k)
        opt = O.Optimizer("Adam")
        bs = t["batch_size"]
        steps = 4608 // bs
        sched = CyclicalLearningRate(2.5e-4, 2.5e-3, 2 * steps)
        it = 0
        for epoch in range(2):
            gen.manual_seed(4 + 1000 * k + epoch)
            perm = torch.randperm(4608, device="cuda", generator=gen).cpu().numpy()
            losses = []
            for s in range(steps):
                idx = perm[s * bs:(s + 1) * bs]
                ws, l, _ = O.train_step(ws, opt, x[idx], y[idx], cfg, sched(it), bf16=True)
                losses.append(l)
                it += 1
            pv = O.forward(ws, xv, cfg, bf16=True)
            assert res[k]["history"]["loss"][epoch] == pytest.approx(float(np.mean(losses)), rel=2e-2), (k, epoch)
            assert res[k]["history"]["val_loss"][epoch] == pytest.approx(float(np.mean((pv - yv) ** 2)), rel=1e-2 if epoch == 0 else 2e-2), (k, epoch)
