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
