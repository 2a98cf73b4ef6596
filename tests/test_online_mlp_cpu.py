"""The online-testing MLP oracle against vectors the REFERENCE produced (tests/golden/online_mlp_golden.npz, made by
tests/golden/make_online_mlp_golden.py from MLP_v2rh/training/mlp.py + torch losses / autograd / torch.optim.Adam):
prediction, loss, every gradient tensor, and five Adam steps.  float32 vs float32: tolerances are accumulation order."""
import os

import numpy as np
import pytest

from oracle import online_mlp_oracle as OO
from online_mlp_inputs import CASES, LR, batches, init_state

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "online_mlp_golden.npz"))


@pytest.mark.parametrize("name", list(CASES))
def test_forward_loss_and_gradients_match_reference(name):
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    pairs = OO.from_state_dict(init_state(name))
    keep = OO.keep_mask(n_out, prune, lev)
    x, y = batches(name)[0]
    lval, grads, pred = OO.loss_and_grads(pairs, x, y, keep, loss)
    np.testing.assert_allclose(pred, GOLD[f"{name}/pred"], rtol=0, atol=2e-6 * np.abs(GOLD[f"{name}/pred"]).max())
    if prune:
        assert not pred[:, 60:60 + lev].any() and not pred[:, 240:240 + lev].any()
    assert (pred[:, -8:] >= 0).all()
    assert abs(lval - float(GOLD[f"{name}/loss"])) <= 1e-6 * abs(float(GOLD[f"{name}/loss"]))
    gsd = OO.to_state_dict(grads)
    for k, g in gsd.items():
        ref = GOLD[f"{name}/grad/{k}"]
        assert g.shape == ref.shape
        np.testing.assert_allclose(g, ref, rtol=0, atol=3e-6 * max(np.abs(ref).max(), 1e-12), err_msg=k)
        if prune and k.startswith("final_linear"):          # pruned columns pass no gradient
            assert not np.take(g, range(60, 60 + lev), axis=0).any()


@pytest.mark.parametrize("name", list(CASES))
def test_five_adam_steps_match_reference(name):
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    pairs = OO.from_state_dict(init_state(name))
    keep = OO.keep_mask(n_out, prune, lev)
    opt = OO.TorchAdam(lr=LR)
    losses = []
    for x, y in batches(name):
        lval, grads, _ = OO.loss_and_grads(pairs, x, y, keep, loss)
        pairs = opt.apply(pairs, grads)
        losses.append(lval)
    np.testing.assert_allclose(losses, GOLD[f"{name}/losses"], rtol=2e-5)
    sd = OO.to_state_dict(pairs)
    for k in [f for f in GOLD.files if f.startswith(f"{name}/after5/")]:
        ref = GOLD[k]
        # Adam's first steps move every weight by ~lr whatever the gradient's size: compare at a fraction of that
        np.testing.assert_allclose(sd[k.split("/after5/")[1]], ref, rtol=0, atol=0.02 * LR, err_msg=k)


def test_state_dict_round_trip_and_keep_mask():
    sd = init_state("huber_prune15")
    back = OO.to_state_dict(OO.from_state_dict(sd))
    assert list(back) == list(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)
    keep = OO.keep_mask(368, True, 12)
    assert keep.sum() == 368 - 48 and not keep[60:72].any() and keep[72] == 1 and not keep[240:252].any() and keep[-8:].all()
    assert OO.keep_mask(368, False, 15).all()


def test_oracle_dropout_mask_and_gradients():
    """Training-mode dropout of the oracle (the mask the HIP engine shares): Bernoulli law of the counter hash, inverted
    scaling, eval mode untouched, and the hand-derived backward against central differences of its own forward loss."""
    keep = OO.dropout_keep_mlp(seed=7, step=3, layer=1, rows=4096, width=256, rate=0.2)
    assert abs(keep.mean() - 0.8) < 3e-3
    assert not np.array_equal(keep, OO.dropout_keep_mlp(7, 4, 1, 4096, 256, 0.2))          # new step, new mask
    assert not np.array_equal(keep, OO.dropout_keep_mlp(7, 3, 2, 4096, 256, 0.2))          # per layer
    assert np.array_equal(keep[:100], OO.dropout_keep_mlp(7, 3, 1, 100, 256, 0.2))         # a function of (row, column) only
    assert abs(np.corrcoef(keep[:-1, :].ravel(), keep[1:, :].ravel())[0, 1]) < 0.01         # rows are independent
    rng = np.random.default_rng(0)
    dims = [12, 128, 128, 16]
    pairs = [(rng.normal(0, 0.3, (a, b)).astype(np.float32), rng.normal(0, 0.1, b).astype(np.float32)) for a, b in zip(dims[:-1], dims[1:])]
    x = rng.normal(0, 1, (64, 12)).astype(np.float32)
    y = rng.normal(0, 1, (64, 16)).astype(np.float32)
    kp = np.ones(16, np.float32)
    d = (0.3, 11, 0)
    p_eval = OO.forward(pairs, x, kp)
    p_tr, hs = OO.forward(pairs, x, kp, keep_acts=True, dropout=d)
    assert not np.allclose(p_eval, p_tr)
    k0 = OO.dropout_keep_mlp(11, 0, 0, 64, 128, 0.3)
    assert np.all(hs[1][~k0] == 0)                                                          # dropped units are exactly zero
    loss, grads, _ = OO.loss_and_grads(pairs, x, y, kp, "mse", dropout=d)
    for trial in range(3):
        dirs = [(rng.normal(0, 1, w.shape).astype(np.float32), rng.normal(0, 1, b.shape).astype(np.float32)) for w, b in pairs]
        eps = 1e-3
        vals = []
        for sgn in (1, -1):
            pp = [(w + np.float32(sgn * eps) * dw, b + np.float32(sgn * eps) * db) for (w, b), (dw, db) in zip(pairs, dirs)]
            vals.append(OO.loss_value(OO.forward(pp, x, kp, dropout=d), y, "mse"))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float(sum((gw.astype(np.float64) * dw).sum() + (gb.astype(np.float64) * db).sum() for (gw, gb), (dw, db) in zip(grads, dirs)))
        assert abs(fd - an) <= 2e-2 * abs(an) + 1e-6, (trial, fd, an)
