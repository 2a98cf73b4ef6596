"""Online-testing MLP on the HIP engine (through the C ABI) against
  (a) vectors the REFERENCE produced (tests/golden/online_mlp_golden.npz: MLP_v2rh/training/mlp.py + torch losses,
      autograd and torch.optim.Adam run in the build container) - bf16-contraction tolerance, and
  (b) the golden-pinned oracle with the engine's bf16 rounding points emulated - accumulation-order tolerance.
Tolerances (same as tests/test_mlp_gpu.py): vs fp32 reference predictions max|d| <= 3e-2 max|ref|, loss 2 %;
vs emulating oracle predictions 2e-3, gradients ||d||/||ref|| <= 5e-3; gradients vs the fp32 reference 15 % in norm
(bf16 operands through up to four layers; huber / mae gradients are sign-like, so a rounding flips whole entries)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import online_mlp_oracle as OO  # noqa: E402
from online_mlp_inputs import CASES, LR, batches, init_state  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "online_mlp_golden.npz"))
PATHS = {"chain": 0, "per_layer": 2}            # CS_FLAG_NO_CHAIN = 2: one GEMM launch per layer


@pytest.fixture(scope="module")
def OM():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd import online_mlp
    return online_mlp


def make(OM, name, flags=0):
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    m = OM.MLP(n_in, n_out, hidden, len(hidden), output_prune=prune, strato_lev_out=lev, loss=loss, max_batch=256, seed=None, flags=flags)
    m.load_state_dict(init_state(name))
    return m


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("name", list(CASES))
def test_forward_loss_gradients(OM, name, path):
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    m = make(OM, name, PATHS[path])
    x, y = batches(name)[0]
    pred = m.forward(x, as_numpy=True)
    gp = GOLD[f"{name}/pred"]
    assert np.abs(pred - gp).max() <= 3e-2 * np.abs(gp).max()
    if prune:
        assert not pred[:, 60:60 + lev].any() and not pred[:, 240:240 + lev].any()      # exactly zero
    assert (pred[:, -8:] >= 0).all()
    pairs = OO.from_state_dict(init_state(name))
    keep = OO.keep_mask(n_out, prune, lev)
    ol, og, op = OO.loss_and_grads(pairs, x, y, keep, loss, bf16=True)
    assert np.abs(pred - op).max() <= 2e-3 * np.abs(op).max()
    lv = m.loss_grads(x, y)
    assert abs(lv - float(GOLD[f"{name}/loss"])) <= 2e-2 * float(GOLD[f"{name}/loss"])
    assert abs(lv - ol) <= 2e-3 * ol
    g = m.gradients()
    osd = OO.to_state_dict(og)
    for k in g:
        assert g[k].shape == GOLD[f"{name}/grad/{k}"].shape
        assert rel(g[k], osd[k]) <= 5e-3, (k, rel(g[k], osd[k]))
        assert rel(g[k], GOLD[f"{name}/grad/{k}"]) <= 0.15, (k, rel(g[k], GOLD[f"{name}/grad/{k}"]))   # bf16 operands, up to 4 layers deep
    if prune:
        assert not g["final_linear.weight"][60:60 + lev].any() and not g["final_linear.bias"][180:180 + lev].any()


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("name", list(CASES))
def test_five_adam_steps(OM, name, path):
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    m = make(OM, name, PATHS[path])
    pairs = OO.from_state_dict(init_state(name))
    keep = OO.keep_mask(n_out, prune, lev)
    opt = OO.TorchAdam(lr=LR)
    losses, olosses = [], []
    for x, y in batches(name):
        losses.append(m.train_step(x, y, LR))
        ol, og, _ = OO.loss_and_grads(pairs, x, y, keep, loss, bf16=True)
        pairs = opt.apply(pairs, og)
        olosses.append(ol)
    np.testing.assert_allclose(losses, GOLD[f"{name}/losses"], rtol=2e-2)
    np.testing.assert_allclose(losses, olosses, rtol=3e-3)
    sd, osd = m.state_dict(), OO.to_state_dict(pairs)
    init = init_state(name)
    for k in sd:
        # total movement after 5 steps is <= 5*lr per weight: compare the MOVEMENT with the oracle's
        mv, omv = sd[k] - init[k], osd[k] - init[k]
        assert rel(mv, omv) <= 5e-3, (k, rel(mv, omv))       # measured 1e-6 (profiles/r05_test_margins.json; the bar was 5e-2 until round 5)
    from conftest import record_margin
    for k in sd:
        record_margin(f"online_{name}_movement_vs_oracle_rel", rel(sd[k] - init[k], osd[k] - init[k]))
    for k in [f for f in GOLD.files if f.startswith(f"{name}/after5/")]:
        kk = k.split("/after5/")[1]
        record_margin(f"online_{name}_movement_vs_reference_rel", rel(sd[kk] - init[kk], GOLD[k] - init[kk]))
        # Adam's first steps are ~lr*sign(g): a bf16-sized difference in a near-zero gradient flips a whole step.  Measured against the
        # reference's fp32 vectors: 0.054 / 0.091 / 0.130 over the three configurations (profiles/r05_test_margins.json, r06: the same); bar 0.17 (0.2 in round 5, 0.35 before)
        assert rel(sd[kk] - init[kk], GOLD[k] - init[kk]) <= 0.17, (kk, rel(sd[kk] - init[kk], GOLD[k] - init[kk]))


@pytest.mark.parametrize("flags", [0, 2])
def test_direct_head_on_the_tuned_chain_kernels(OM, flags):
    """128 outputs and 128/256/512-wide layers run on the layer-chain kernels (k_chain): huber + a keep mask there."""
    from climsim_amd.mlp import MLPEmulator
    rs = np.random.RandomState(5)
    dims = [124, 256, 512, 128]
    keep = np.ones(128, np.float32)
    keep[60:72] = 0
    pairs = [((rs.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
              (rs.standard_normal(dims[i + 1]) * 0.05).astype(np.float32)) for i in range(3)]
    x = (rs.standard_normal((200, 124)) * 0.5).astype(np.float32)
    y = (rs.standard_normal((200, 128)) * 1.5).astype(np.float32)
    m = MLPEmulator(units=(256, 512), activation="relu", optimizer="AdamTorch", max_batch=256, seed=None, epsilon=1e-8,
                    direct_head=True, loss="huber", output_keep=keep, flags=flags)
    w, b = pairs[-1]
    m.set_weights([pairs[0][0], pairs[0][1], pairs[1][0], pairs[1][1], w[:, :120].copy(), b[:120].copy(), w[:, 120:].copy(), b[120:].copy()])
    ol, og, op = OO.loss_and_grads(pairs, x, y, keep, "huber", bf16=True)
    pred = m.predict(x)
    assert np.abs(pred - op).max() <= 2e-3 * np.abs(op).max() and not pred[:, 60:72].any()
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    sums = m.loss_grads(xd, yd).cpu().numpy()
    assert abs(sums[0] / (128 * 200) - ol) <= 2e-3 * ol
    g = m.get_gradients(1.0 / (128 * 200))
    flat_o = [og[0][0], og[0][1], og[1][0], og[1][1], og[2][0][:, :120], og[2][1][:120], og[2][0][:, 120:], og[2][1][120:]]
    for a, o in zip(g, flat_o):
        assert rel(a, o) <= 5e-3
    opt = OO.TorchAdam(lr=LR)
    for _ in range(4):
        m.train_on_batch(xd, yd, LR)
        _, og, _ = OO.loss_and_grads(pairs, x, y, keep, "huber", bf16=True)
        pairs = opt.apply(pairs, og)
    ws = m.get_weights()
    assert rel(ws[0], pairs[0][0]) <= 3e-3
    assert rel(ws[3], pairs[1][1]) <= 2e-2


def test_state_dict_fit_and_errors(OM):
    m = make(OM, "huber_prune15")
    sd = init_state("huber_prune15")
    back = m.state_dict()
    assert list(back) == list(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)        # fp32 master weights: exact
    rs = np.random.RandomState(0)
    x = (rs.standard_normal((2048, 64)) * 0.5).astype(np.float32)
    A = (rs.standard_normal((64, 368)) / 8).astype(np.float32)
    y = np.tanh(x @ A).astype(np.float32)
    y[:, -8:] = np.abs(y[:, -8:])
    h = m.fit(x, y, batch_size=256, epochs=6, learning_rate=4e-3, scheduler=OM.StepLR(4e-3, 2, 0.5), validation_data=(x[:512], y[:512]))
    assert h["loss"][-1] < 0.9 * h["loss"][0] and h["val_loss"][-1] < h["val_loss"][0]
    assert all(a > b for a, b in zip(h["loss"], h["loss"][1:]))
    assert h["lr"] == [4e-3, 4e-3, 2e-3, 2e-3, 1e-3, 1e-3]
    assert abs(m.evaluate(x[:512], y[:512]) - h["val_loss"][-1]) < 1e-6
    md = OM.MLP(64, 368, 128, 2, dropout=0.1, max_batch=256)          # dropout is built (test_training_mode_dropout_follows_the_oracle)
    assert md.dropout == 0.1
    with pytest.raises(AssertionError):
        OM.MLP(64, 368, [128], 2)
    with pytest.raises(ValueError):
        OM.MLP(64, 128, [128], 1, output_prune=True)


def test_export_wrapper_matches_device_pipeline(OM, tmp_path):
    """v2_nn_wrapper.ipynb cell 5: normalise -> model -> zero pruned outputs -> / out_scale, as a TorchScript file."""
    m = make(OM, "v2rh_mse_prune12")
    rs = np.random.RandomState(3)
    sub = rs.standard_normal(557).astype(np.float32)
    div = (0.5 + rs.random_sample(557)).astype(np.float32)
    div[7] = 0.0                                                     # x/0 -> inf/nan -> 0
    scale = (0.5 + rs.random_sample(368)).astype(np.float32)
    raw = (rs.standard_normal((64, 557)) * 2).astype(np.float32)
    mod = m.export_wrapper(str(tmp_path / "wrapper.pt"), sub, div, scale)
    loaded = torch.jit.load(str(tmp_path / "wrapper.pt"))
    out = loaded(torch.from_numpy(raw.copy())).numpy()
    assert np.array_equal(out, mod(torch.from_numpy(raw.copy())).numpy())
    with np.errstate(divide="ignore", invalid="ignore"):
        xn = (raw - sub) / div
    xn[~np.isfinite(xn)] = 0
    xn[:, 60:120] = np.clip(xn[:, 60:120], 0, 1.2)
    y = m.forward(xn, as_numpy=True)
    for a, b in ((60, 75), (120, 148), (180, 195), (240, 255), (300, 315)):
        y[:, a:b] = 0
    ref = y / scale
    assert np.abs(out - ref).max() <= 3e-2 * np.abs(ref).max()       # fp32 TorchScript vs bf16 engine
    assert not out[:, 120:148].any() and not out[:, 300:315].any()


@pytest.mark.parametrize("name,rate", [("v2rh_mse_prune12", 0.1), ("huber_prune15", 0.25)])
def test_training_mode_dropout_follows_the_oracle(OM, name, rate):
    """`MLP(..., dropout=p)` (mlp.py:39-44: Sequential(Linear, Dropout) -> relu): the training pass draws the mask the oracle
    draws (shared counter hash; torch's own stream is not reproducible), so loss and every gradient tensor are held to the
    bf16-emulating oracle at the usual tolerances; prediction / evaluation stay in eval mode; every optimiser step draws a
    new mask."""
    n_in, n_out, hidden, prune, lev, loss, nb = CASES[name]
    m = OM.MLP(n_in, n_out, hidden, len(hidden), dropout=rate, output_prune=prune, strato_lev_out=lev, loss=loss, max_batch=256, seed=None,
               dropout_seed=1234567890123)
    m.load_state_dict(init_state(name))
    x, y = batches(name)[0]
    pairs = OO.from_state_dict(init_state(name))
    keep = OO.keep_mask(n_out, prune, lev)
    # eval mode: identical to the model without dropout
    op = OO.forward(pairs, x, keep, bf16=True)
    pred = m.forward(x, as_numpy=True)
    assert np.abs(pred - op).max() <= 2e-3 * np.abs(op).max()
    ev = m.evaluate(x, y)
    assert abs(ev - OO.loss_value(op, y, loss)) <= 2e-3 * ev
    # training mode, step 0
    ol, og, _ = OO.loss_and_grads(pairs, x, y, keep, loss, bf16=True, dropout=(rate, 1234567890123, 0))
    ol_eval, _, _ = OO.loss_and_grads(pairs, x, y, keep, loss, bf16=True)
    lv = m.loss_grads(x, y)
    assert abs(lv - ol) <= 2e-3 * ol and abs(ol - ol_eval) > 1e-3 * ol_eval      # the mask matters, and it is the oracle's
    g, osd = m.gradients(), OO.to_state_dict(og)
    for k in g:
        assert rel(g[k], osd[k]) <= 5e-3, (k, rel(g[k], osd[k]))
    assert m.loss_grads(x, y) == pytest.approx(lv, rel=1e-5)                       # same step -> same mask
    # optimiser steps: the mask key follows the step counter
    opt = OO.TorchAdam(lr=LR)
    for step in range(3):
        l_ref, grads, _ = OO.loss_and_grads(pairs, x, y, keep, loss, bf16=True, dropout=(rate, 1234567890123, step))
        pairs = opt.apply(pairs, grads)
        l_eng = m.train_step(x, y, LR)
        assert abs(l_eng - l_ref) <= 2e-2 * l_ref, (step, l_eng, l_ref)
    with pytest.raises(ValueError):
        OM.MLP(n_in, n_out, hidden, len(hidden), dropout=1.0, max_batch=64)
