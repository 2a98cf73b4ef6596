"""oracle/mlp_oracle.py - the oracle of the BENCHMARKED kernels - against vectors the REFERENCE computed.

tests/golden/hot_mlp_golden.npz is made by tests/golden/make_online_mlp_golden.py: the reference's torch `MLP`
(online_testing/baseline_models/MLP_v2rh/training/mlp.py:28-67) instantiated as 124 -> [512]*5 + [128] -> 128 (= the cfg-MLP of
step2_retrain.py:95-126 with activation 'relu': the two heads are one 128 x 128 layer with ReLU on columns 120..127) at batch
8192 and as the published 768-640-512-640-640 model at batch 3072, `nn.MSELoss`, autograd and five `torch.optim.Adam` steps
(train_mlp_h5loader.py:210-211,226-236).  float32 against float32: the tolerances are accumulation order (3e-6 of a tensor's
largest entry; Adam's movement as in test_online_mlp_cpu.py).  This pins topology, head fusion, MSE, the hand-derived backward and
the torch-Adam rule of the hot-path oracle; LeakyReLU / ELU epilogues and the Keras / tfa optimiser rules stay restated."""
import os

import numpy as np
import pytest

from oracle import mlp_oracle as O
from online_mlp_inputs import HOT_CASES, LR, hot_batches, hot_init_state, hot_pred_rows, hot_summary

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hot_mlp_golden.npz"))


def keras_list(sd, n_hidden):
    """torch state_dict of the reference MLP -> the Keras-ordered list of the baseline model (heads split 120 / 8)."""
    ws = []
    for i in range(n_hidden):
        ws += [sd[f"linears.{i}.0.weight"].T.copy(), sd[f"linears.{i}.0.bias"].copy()]
    w, b = sd["final_linear.weight"].T, sd["final_linear.bias"]
    return ws + [w[:, :120].copy(), b[:120].copy(), w[:, 120:].copy(), b[120:].copy()]


def state_dict(ws, n_hidden):
    sd = {}
    for i in range(n_hidden):
        sd[f"linears.{i}.0.weight"], sd[f"linears.{i}.0.bias"] = ws[2 * i].T, ws[2 * i + 1]
    sd["final_linear.weight"] = np.concatenate([ws[-4], ws[-2]], axis=1).T
    sd["final_linear.bias"] = np.concatenate([ws[-3], ws[-1]])
    return sd


def cfg_of(name):
    n_in, n_out, hidden, _, _ = HOT_CASES[name]
    return O.MLPConfig(n_in=n_in, hidden=tuple(hidden[:-1]), act="relu")


def summary_errors(name, prefix, key, t):
    """Deviations of a weight-shaped tensor from what the fixture keeps of the reference's one: relative norm difference,
    projections relative to the norm, corner / strided-sample entries relative to the tensor's rms entry."""
    got = hot_summary(name, key, t)
    fro = float(GOLD[f"{prefix}/{key}/fro"])
    rms = fro / np.sqrt(np.asarray(t).size)
    return {"fro": abs(got["fro"] - fro) / fro, "proj": float(np.abs(got["proj"] - GOLD[f"{prefix}/{key}/proj"]).max() / fro),
            "entry": float(max(np.abs(got[f] - GOLD[f"{prefix}/{key}/{f}"]).max() for f in ("corner", "sample")) / rms)}


# Accumulation order (float32 sums of the reference against float64 sums rounded once): what every tensor meets unless a ReLU mask differs.
TIGHT = {"fro": 3e-6, "proj": 3e-6, "entry": 1e-5, "bias": 3e-6}
# A pre-activation within float32 rounding of zero (|z| <~ 1e-7 sum|terms|: 48 candidates among the 22 M of the batch-8192 case) gets
# its ReLU mask from the summation order: one (row, unit) of one layer then passes or blocks its gradient, which changes that
# layer's gradients and those of the layers BELOW it by one row's share - measured 2e-4 of a tensor's norm, 9e-4 of its rms entry.
LOOSE = {"fro": 3e-6, "proj": 1e-3, "entry": 5e-3, "bias": 2e-4}


@pytest.mark.parametrize("name", list(HOT_CASES))
def test_forward_loss_and_every_gradient_match_the_reference(name):
    n_in, n_out, hidden, loss, nb = HOT_CASES[name]
    cfg = cfg_of(name)
    assert cfg.dims == [n_in, *hidden]
    ws = keras_list(hot_init_state(name), len(hidden))
    x, y = hot_batches(name)[0]
    lval, mae, grads, pred = O.loss_and_grads(ws, x, y, cfg, bf16=False)
    ref_rows = GOLD[f"{name}/pred_rows"]
    assert np.abs(pred[hot_pred_rows(name)] - ref_rows).max() <= 3e-6 * float(GOLD[f"{name}/pred_absmax"])
    assert abs(np.abs(pred).max() - float(GOLD[f"{name}/pred_absmax"])) <= 3e-6 * float(GOLD[f"{name}/pred_absmax"])
    assert (pred[:, 120:] >= 0).all()
    assert abs(lval - float(GOLD[f"{name}/loss"])) <= 1e-6 * float(GOLD[f"{name}/loss"])
    assert abs(mae - float(GOLD[f"{name}/mae"])) <= 1e-6 * float(GOLD[f"{name}/mae"])
    gsd = state_dict(grads, len(hidden))
    tight = []                                       # per layer, bottom to top: does every gradient meet the accumulation-order bar?
    for li in range(len(hidden) + 1):
        key = f"linears.{li}.0" if li < len(hidden) else "final_linear"
        gb, rb = gsd[key + ".bias"], GOLD[f"{name}/grad/{key}.bias"]
        assert gb.shape == rb.shape
        errs = summary_errors(name, f"{name}/grad", key + ".weight", gsd[key + ".weight"])
        errs["bias"] = float(np.abs(gb - rb).max() / np.abs(rb).max())
        for f, e in errs.items():
            assert e <= LOOSE[f], (key, f, e)
        tight.append(all(e <= TIGHT[f] for f, e in errs.items()))
    # a differing mask at layer k touches layers <= k only: the layers that miss the tight bar are a run from the bottom, and the
    # three layers on top (heads, 128-wide, last trunk layer) carry none in either case
    assert tight == sorted(tight), tight
    assert all(tight[-3:]), tight


@pytest.mark.parametrize("name", list(HOT_CASES))
def test_five_torch_adam_steps_match_the_reference(name):
    """Five steps of the reference's optimiser.  Adam's first steps move every weight by ~lr * sign-like(g): entries whose gradient
    is near zero amplify any difference (measured: projections of the movement 1.2e-3 of its norm, entries 6e-3 of the rms
    movement where a ReLU mask differed, 1e-5 / 2e-4 where none did)."""
    n_in, n_out, hidden, loss, nb = HOT_CASES[name]
    cfg = cfg_of(name)
    init = hot_init_state(name)
    ws = keras_list(init, len(hidden))
    opt = O.Optimizer(kind="AdamTorch", eps=1e-8)
    losses = []
    for x, y in hot_batches(name):
        ws, lval, _ = O.train_step(ws, opt, x, y, cfg, LR)
        losses.append(lval)
    np.testing.assert_allclose(losses, GOLD[f"{name}/losses"], rtol=2e-5)
    sd = state_dict(ws, len(hidden))
    for k, v in sd.items():
        if k.endswith("bias"):
            np.testing.assert_allclose(v, GOLD[f"{name}/after5/{k}"], rtol=0, atol=0.02 * LR, err_msg=k)
            mv, rmv = v - init[k], GOLD[f"{name}/after5/{k}"] - init[k]
            assert np.linalg.norm(mv - rmv) <= 1e-3 * np.linalg.norm(rmv), k
        else:
            errs = summary_errors(name, f"{name}/moved5", k, v.astype(np.float64) - init[k])
            assert errs["fro"] <= 2e-5 and errs["proj"] <= 5e-3 and errs["entry"] <= 2e-2, (k, errs)


def test_known_answers_of_the_two_topologies():
    """step1_results.csv:170 / FLOP_calculation.ipynb: the published model's parameter count and forward FLOPs."""
    pub = cfg_of("pub_mlp_b3072")
    assert pub.n_params() == 1_753_472 and pub.fwd_flops() == 3_503_488
    assert cfg_of("cfg_mlp_b8192").train_flops() == 7_036_928
