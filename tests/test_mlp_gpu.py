"""GPU parity tests: the HIP engine (through the C-ABI, via climsim_amd.mlp) against the CPU oracle
on the same seeded inputs, plus size-independent properties at the BASELINE.json config size.

Tolerances (stated once, used below):
  * bf16-emulating oracle (same rounding points, float64 accumulation): the only differences are the
    fp32 accumulation order and rare 1-ulp bf16 flips of activations -> activations/predictions
    max|d| <= 2e-3*max|ref|; gradients ||d||/||ref|| <= 5e-3 per tensor.
  * pure-fp32 oracle (what Keras computes, up to TF32): bf16 operand rounding (2^-9 relative per
    operand) -> predictions max|d| <= 3e-2*max|ref|, loss within 2 %.
  * fp32 elementwise kernels (normalise, optimiser): <= 2 float32 ulp (1e-6 relative).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import mlp_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd import mlp
    return mlp


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / (np.linalg.norm(b) + 1e-30))


def make_model(M, units, act="leakyrelu", opt="Adam", max_batch=1024, seed=3, flags=0, bias_scale=0.05):
    m = M.MLPEmulator(units=units, activation=act, optimizer=opt, max_batch=max_batch, seed=None, flags=flags)
    cfg = O.MLPConfig(hidden=tuple(units), act=act)
    ws = O.glorot_init(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for i in range(1, len(ws), 2):           # non-zero biases so the bias path is exercised
        ws[i] = rng.normal(0, bias_scale, ws[i].shape).astype(np.float32)
    m.set_weights(ws)
    return m, cfg, ws


# engine flags: 0 = layer-chain kernels (tile by batch: 32 rows at these sizes), 2 = one GEMM per layer,
# 4/8/16 = chain with 64/128/32-row tiles
@pytest.mark.parametrize("flags", [0, 2, 4, 8, 66])      # 66 = per-layer path on the first GEMM kernels
@pytest.mark.parametrize("act", ["relu", "elu", "leakyrelu"])
@pytest.mark.parametrize("n", [1, 200, 384])
def test_forward_matches_oracle(M, act, n, flags):
    m, cfg, ws = make_model(M, (128, 256, 512), act=act, flags=flags)
    x, _ = O.synth_columns(n, seed=11)
    got = m.predict(x)
    ref16 = O.forward(ws, x, cfg, bf16=True)
    ref32 = O.forward(ws, x, cfg, bf16=False)
    assert got.shape == (n, 128) and got.dtype == np.float32
    assert np.max(np.abs(got - ref16)) <= 2e-3 * np.max(np.abs(ref16))
    assert np.max(np.abs(got - ref32)) <= 3e-2 * np.max(np.abs(ref32))
    assert np.all(got[:, 120:] >= 0)                      # relu head


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 4, 8, 36, 66, 256])  # +1 = CS_FLAG_NO_TR_READ; 36 = 64-row forward + 32-row backward; 66 = first GEMM kernels;
# 0 runs forward + backward chain in ONE launch (k_chain_fb, 32-row tiles), 256 = CS_FLAG_NO_CHAIN_FB: two launches
@pytest.mark.parametrize("act,n,units", [("leakyrelu", 300, (256, 128, 512)), ("relu", 128, (512, 512)),
                                         ("elu", 1000, (256, 128, 384)),    # 384: wide chain (chainw.h), or per-layer with flag 2
                                         ("leakyrelu", 200, (768, 640, 512, 640, 640)),   # the published lot-147/trial_0027 widths
                                         ("relu", 77, (1024, 896)),
                                         ("relu", 90, (256, 128, 384, 128, 256, 128, 128, 384, 256, 128))])   # 10 hidden layers
def test_loss_and_gradients_match_oracle(M, act, n, units, flags):
    m, cfg, ws = make_model(M, units, act=act, flags=flags)
    x, y = O.synth_columns(n, seed=7)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.gradient_tensor()
    loss = m.loss_grads(xd, yd).cpu().numpy().astype(np.float64)
    ref_loss, ref_mae, ref_g, _ = O.loss_and_grads(ws, x, y, cfg, bf16=True)
    assert loss[0] / (128 * n) == pytest.approx(ref_loss, rel=2e-3)
    assert loss[1] / (128 * n) == pytest.approx(ref_mae, rel=2e-3)
    got = m.get_gradients(1.0 / (128 * n))
    assert len(got) == len(ref_g)
    for i, (g, r) in enumerate(zip(got, ref_g)):
        assert g.shape == r.shape
        assert rel(g, r) <= 5e-3, (i, rel(g, r))
    l32, _, g32, _ = O.loss_and_grads(ws, x, y, cfg, bf16=False)
    assert loss[0] / (128 * n) == pytest.approx(l32, rel=2e-2)
    # bf16 operand rounding accumulates with depth (5 hidden layers: measured 6.6e-2; 10: 1.03e-1, identical for the
    # chain and the per-layer kernels - the bf16-aware oracle above is the parity check, this one bounds the precision)
    tol32 = 6e-2 if len(units) <= 3 else 9e-2 if len(units) <= 7 else 1.4e-1
    for g, r in zip(got, g32):
        assert rel(g, r) <= tol32


@pytest.mark.parametrize("opt", ["Adam", "RAdam", "RMSprop", "SGD"])
def test_optimizer_kernel_matches_oracle(M, opt):
    m, cfg, ws = make_model(M, (128, 128), opt=opt)
    g = m.gradient_tensor()
    ref = O.Optimizer(opt)
    w = [a.copy() for a in ws]
    rng = np.random.default_rng(5)
    for step in range(7):                                  # crosses RAdam's sma_t >= 5 switch (t = 6)
        grads = [rng.normal(0, 1e-3, a.shape).astype(np.float32) for a in ws]
        # internal flat layout = Keras order with the two heads fused column-wise
        fused = O.fuse_heads(grads)
        flat = np.concatenate([np.concatenate([gw.ravel(), gb.ravel()]) for gw, gb in fused])
        g.copy_(torch.from_numpy(flat))
        lr = 1e-3 * (1 + step)
        m.apply_gradients(lr, 1.0)
        w = ref.apply(w, grads, lr)
    got = m.get_weights()
    for a, b in zip(got, w):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-8)   # ~2 ulp of an lr-sized update
    gm, gv, it = m.get_optimizer_state()
    assert it == 7
    if opt in ("Adam", "RAdam"):
        for a, b in zip(gm, ref.m):
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-6 * np.max(np.abs(b)))   # fma contraction: few ulp
    if opt != "SGD":
        for a, b in zip(gv, ref.v):
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-6 * np.max(np.abs(b)))


def test_training_curve_tracks_oracle(M):
    units = (128, 128)
    m, cfg, ws = make_model(M, units, bias_scale=0.0)
    x, y = O.synth_columns(512, seed=21)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    opt = O.Optimizer("Adam")
    w = ws
    sched = M.CyclicalLearningRate(step_size=4)
    losses_g, losses_r = [], []
    for it in range(12):
        lr = sched(it)
        assert lr == pytest.approx(O.cyclical_lr(it, step_size=4))
        lg = m.train_on_batch(xd, yd, lr).cpu().numpy()[0] / (128 * 512)
        w, lr_, _ = O.train_step(w, opt, x, y, cfg, lr, bf16=True)
        losses_g.append(lg)
        losses_r.append(lr_)
    np.testing.assert_allclose(losses_g, losses_r, rtol=2e-2)
    assert losses_g[-1] < 0.7 * losses_g[0]
    got = m.get_weights()
    # Adam normalises tiny gradients, so individual weights may differ by O(lr) where g ~ 0;
    # compare in aggregate
    worst = max(rel(a - w0, b - w0) for a, b, w0 in zip(got, w, ws))
    print("worst relative difference of the 12-step weight movement:", worst)
    assert worst <= 2e-2                      # measured 2.7e-3 (a bar of 0.15 stood here while the measured value was this)


def test_weights_and_optimizer_state_roundtrip(M, tmp_path):
    m, cfg, ws = make_model(M, (128, 256))
    for a, b in zip(m.get_weights(), ws):
        np.testing.assert_array_equal(a, b)
    assert m.count_params() == cfg.n_params() == sum(a.size for a in ws)
    x, y = O.synth_columns(256, seed=2)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for _ in range(3):
        m.train_on_batch(xd, yd, 1e-3)
    p = str(tmp_path / "ckpt.npz")
    m.save_weights(p)
    pred = m.predict(x)
    m2 = M.MLPEmulator(units=(128, 256), seed=None, max_batch=1024)
    m2.load_weights(p)
    np.testing.assert_array_equal(m2.predict(x), pred)
    assert m2.iterations == 3
    l1 = m.train_on_batch(xd, yd, 1e-3).cpu().numpy()
    l2 = m2.train_on_batch(xd, yd, 1e-3).cpu().numpy()
    np.testing.assert_allclose(l1, l2, rtol=1e-5)
    with pytest.raises(ValueError):
        m.set_weights(ws[:-1])
    # the same through a Keras-layout .h5 checkpoint (climsim_amd/keras_h5.py; step2_retrain.py:253-261)
    ph = str(tmp_path / "ckpt.h5")
    m.save_weights(ph)
    m3 = M.MLPEmulator(units=(128, 256), seed=None, max_batch=1024)
    m3.load_weights(ph)
    for a, b in zip(m3.get_weights(), m.get_weights()):
        np.testing.assert_array_equal(a, b)
    assert m3.iterations == m.iterations
    np.testing.assert_allclose(m3.train_on_batch(xd, yd, 1e-3).cpu().numpy(), m.train_on_batch(xd, yd, 1e-3).cpu().numpy(), rtol=1e-5)


def test_in_kernel_normalise_gather_and_inf_nan_rule(M, lowres_assets):
    from climsim_amd.data_utils import data_utils
    import copy
    grid, *sets = lowres_assets
    d = data_utils(copy.copy(grid), *sets, ml_backend="pytorch")
    d.set_to_v1_vars()
    sub, div, _ = d.save_norm()
    m, cfg, ws = make_model(M, (128, 128))
    m.set_norm(sub, div)
    xn, _ = O.synth_columns(300, seed=13)
    raw = (xn.astype(np.float64) * div + sub).astype(np.float32)        # un-normalised columns
    raw[5, 3] = np.inf
    raw[6, 70] = np.nan
    idx = np.random.default_rng(0).permutation(300)[:200].astype(np.int64)
    xd = torch.from_numpy(raw).cuda()
    out = torch.empty((200, 128), dtype=torch.float32, device="cuda")
    m.forward_batch(xd, yhat=out, row_idx=torch.from_numpy(idx).cuda(), normalise=True)
    ref_in = O.normalise(raw[idx], sub, div)
    assert np.all(np.isfinite(ref_in))
    ref = O.forward(ws, ref_in, cfg, bf16=True)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 2e-3 * np.max(np.abs(ref))
    # stand-alone loader-path kernel: bit-exact fp32 (IEEE subtract + divide)
    from climsim_amd import _lib
    import ctypes as C
    lib = _lib.load()
    o2 = torch.empty((200, 124), dtype=torch.float32, device="cuda")
    sd = torch.from_numpy(sub.astype(np.float32)).cuda()
    dd = torch.from_numpy(div.astype(np.float32)).cuda()
    _lib.check(lib.cs_normalise_rows(C.c_void_p(xd.data_ptr()), C.c_void_p(torch.from_numpy(idx).cuda().data_ptr()), 200, 124,
                                     C.c_void_p(sd.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(o2.data_ptr()), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o2.cpu().numpy(), ref_in)


def test_error_behaviour(M):
    m, *_ = make_model(M, (128, 128), max_batch=256)
    from climsim_amd import _lib
    x = torch.zeros((512, 124), device="cuda")
    with pytest.raises(_lib.EngineError):
        m.train_on_batch(x, torch.zeros((512, 128), device="cuda"), 1e-3)      # n > max_batch: max_batch sizes the training buffers
    m2, *_ = make_model(M, (128, 128), max_batch=256, flags=2)                 # per-layer path: forward stages activations, same bound
    with pytest.raises(_lib.EngineError):
        m2.forward_batch(x, yhat=torch.empty((512, 128), device="cuda"))
    assert m2.lib.cs_mlp_forward_limit(m2._h) == 256 and m.lib.cs_mlp_forward_limit(m._h) >= 1 << 20
    m2.close()
    with pytest.raises(_lib.EngineError):
        m.forward_batch(x[:8], yhat=torch.empty((8, 128), device="cuda"), normalise=True)   # no norm set
    with pytest.raises(ValueError):
        m.predict(np.zeros((4, 100), np.float32))


@pytest.mark.parametrize("units,act,n", [((512,) * 5, "leakyrelu", 70001), ((768, 640, 512, 640, 640), "leakyrelu", 20000), ((256, 128), "elu", 33000)])
def test_prediction_and_evaluation_take_calls_beyond_max_batch(M, units, act, n):
    """Round 4: max_batch sizes the TRAINING buffers; prediction / evaluation on the layer-chain paths keep nothing per row, so one
    cs_mlp_forward call takes far more rows (cs_mlp_forward_limit) and `predict` / `evaluate` default to calls of 65536 (tall tiles fill
    the chip: 390 M columns/s against 200 M in calls of 8192).  Same predictions as calls of max_batch rows (to accumulation order) and
    the same sums; held to the oracle."""
    m, cfg, ws = make_model(M, units, act, max_batch=2048)
    x, y = O.synth_columns(n, seed=5)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    big, small = m.predict(xd, as_numpy=False), m.predict(xd, batch_size=2048, as_numpy=False)
    # taller tiles sum a 128-wide stage's contraction in another order than 32-row tiles (which split it over the wave halves):
    # accumulation-order tolerance, the one the oracle comparison uses
    assert float((big - small).abs().max()) <= 6e-3 * float(small.abs().max()) and rel(big.cpu().numpy(), small.cpu().numpy()) <= 1e-3
    one = torch.empty_like(big)
    m.forward_batch(xd, yhat=one)                                               # ONE call of n rows
    assert float((one - small).abs().max()) <= 6e-3 * float(small.abs().max())
    again = torch.empty_like(big)
    m.forward_batch(xd, yhat=again)
    assert torch.equal(one, again)                                              # and deterministic
    ref = O.forward(ws, x[:512], cfg, bf16=True)
    got = big[:512].cpu().numpy()                                               # the large-size bar of test_mlp_large_gpu.py (deep models, many outputs)
    assert rel(got, ref) <= 1e-3 and np.abs(got - ref).max() <= 6e-3 * np.abs(ref).max()
    eb, es = m.evaluate(xd, yd), m.evaluate(xd, yd, batch_size=2048)
    assert eb["loss"] == pytest.approx(es["loss"], rel=1e-5) and eb["mae"] == pytest.approx(es["mae"], rel=1e-5)
    m.close()


def test_fit_predict_evaluate_api(M, tmp_path):
    m, cfg, ws = make_model(M, (128, 128), max_batch=512)
    x, y = O.synth_columns(4096, seed=1)
    xv, yv = O.synth_columns(1024, seed=2)
    before = m.evaluate(xv, yv)
    hist = m.fit(x, y, batch_size=256, epochs=3, validation_data=(xv, yv),
                 learning_rate=M.CyclicalLearningRate(step_size=2 * 16), csv_log=str(tmp_path / "log.csv"),
                 checkpoint_best=str(tmp_path / "best.npz"), checkpoint_last=str(tmp_path / "last.npz"),
                 early_stopping_patience=8)
    assert len(hist["loss"]) == 3 and hist["val_loss"][-1] < before["loss"]
    assert m.iterations == 3 * 16
    rows = open(tmp_path / "log.csv").read().strip().splitlines()
    # keras.callbacks.CSVLogger layout (baseline_models/ED/model/ED_ClimSIM_1_3.csv:1): epoch + sorted log keys, "NA" where absent
    assert rows[0] == "epoch,accuracy,loss,lr,mae,mse,val_accuracy,val_loss,val_mae,val_mse" and len(rows) == 4
    last = dict(zip(rows[0].split(","), rows[-1].split(",")))
    # round 4: the training-pass `accuracy` column is filled (cs_mlp_set_train_accuracy; "NA" in rounds 1-3)
    assert 0.0 <= float(last["accuracy"]) <= 1.0 and float(last["accuracy"]) == pytest.approx(hist["accuracy"][-1])
    assert float(last["val_loss"]) == pytest.approx(hist["val_loss"][-1])
    ev = m.evaluate(xv, yv, accuracy=True)
    pv = m.predict(xv)
    assert ev["accuracy"] == pytest.approx(float(np.mean(pv.argmax(1) == yv.argmax(1))), abs=1e-12)   # Keras categorical_accuracy
    assert float(last["val_accuracy"]) == pytest.approx(ev["accuracy"])
    after = m.evaluate(xv, yv, batch_size=100)                      # ragged batches
    assert after["loss"] == pytest.approx(hist["val_loss"][-1], rel=1e-4)
    # the training-pass accuracy is Keras' categorical accuracy of the batches the steps predicted: with a zero learning rate
    # (weights stand still) one epoch over the first 1024 rows must count exactly what predict + argmax count on those rows
    h0 = m.fit(x[:1024], y[:1024], batch_size=256, epochs=1, learning_rate=0.0, shuffle=True, train_accuracy=True)
    p0 = m.predict(x[:1024])
    assert h0["accuracy"][0] == pytest.approx(float(np.mean(p0.argmax(1) == y[:1024].argmax(1))), abs=1e-12)
    assert "accuracy" not in m.fit(x[:1024], y[:1024], batch_size=256, epochs=1, learning_rate=0.0)      # off without a CSV log
    pred = m.predict(xv, batch_size=300)
    assert np.mean((pred - yv) ** 2) == pytest.approx(after["mse"], rel=1e-3)
    # a fit that fails before its first step (the CSV log cannot be opened) must not leave the engine with the device pointer of its
    # accuracy counter: the next steps would add to freed memory (round-4 advisor finding).  Afterwards the model trains as before.
    with pytest.raises(OSError):
        m.fit(x[:1024], y[:1024], batch_size=256, epochs=1, csv_log=str(tmp_path / "no_such_dir" / "log.csv"))
    w0 = m.get_weights()
    h1 = m.fit(x[:1024], y[:1024], batch_size=256, epochs=1, learning_rate=0.0)
    assert "accuracy" not in h1 and all(np.array_equal(a, b) for a, b in zip(w0, m.get_weights()))


# ----------------------------------------------------------------------- BASELINE-size properties
@pytest.fixture(scope="module")
def big(M):
    m = M.MLPEmulator(units=(512,) * 5, max_batch=8192, seed=0)
    x, y = O.synth_columns(8192, seed=20230614)
    return m, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), x, y


def test_full_size_row_independence_is_bit_exact(big):
    m, xd, yd, x, y = big
    full = m.predict(xd, as_numpy=False)
    a = m.predict(xd[:3000], as_numpy=False)
    b = m.predict(xd[3000:], as_numpy=False)
    assert torch.equal(full, torch.cat([a, b]))
    perm = torch.randperm(8192, device="cuda")
    out = torch.empty_like(full)
    m.forward_batch(xd, yhat=out, row_idx=perm)
    assert torch.equal(out, full[perm])


def test_full_size_gradient_additivity_and_determinism(big):
    m, xd, yd, x, y = big
    g = m.gradient_tensor()
    l_full = m.loss_grads(xd, yd).clone()
    g_full = g.clone()
    l_again = m.loss_grads(xd, yd).clone()
    assert torch.allclose(l_full, l_again, rtol=1e-5)
    assert float((g - g_full).norm() / g_full.norm()) < 1e-5          # atomics order only
    la = m.loss_grads(xd[:4096], yd[:4096]).clone()
    m.loss_grads(xd[4096:], yd[4096:], accumulate=True)
    assert float((g - g_full).norm() / g_full.norm()) < 1e-5          # sum over micro-batches
    assert torch.allclose(m._loss, l_full, rtol=1e-5) and float(la[0]) < float(l_full[0])
    # against the oracle on a subset that finishes in seconds
    ws = m.get_weights()
    cfg = O.MLPConfig()
    m.loss_grads(xd[:1024], yd[:1024])
    ref_loss, _, ref_g, _ = O.loss_and_grads(ws, x[:1024], y[:1024], cfg, bf16=True)
    assert float(m._loss[0]) / (128 * 1024) == pytest.approx(ref_loss, rel=2e-3)
    for a, b in zip(m.get_gradients(1.0 / (128 * 1024)), ref_g):
        assert rel(a, b) <= 5e-3


def test_full_size_training_reduces_loss(big):
    m, xd, yd, x, y = big
    first = float(m.train_on_batch(xd, yd, 1e-3)[0])
    for _ in range(20):
        last = float(m.train_on_batch(xd, yd, 1e-3)[0])
    assert np.isfinite(last) and last < 0.5 * first


def test_heldout_per_variable_mae_r2_match_cpu_training(M, lowres_assets):
    """BASELINE acceptance ("MAE acceptance", synthetic form): the same model trained for the same 1600
    steps on the same batches (Adam, lr 1e-3 then 1e-4 for the last quarter) by the HIP engine (bf16 operands) and by the
    fp32 torch-CPU restatement of the reference step (oracle/mlp_torch_cpu.py), both scored on a held-out split through
    the evaluation pipeline of data_utils (set_pressure_grid -> output_weighting -> calc_MAE / RMSE / R2 ->
    create_metrics_df; golden-pinned on the CPU).
    Tolerance: MAE and RMSE within 2 % relative for the two 60-level variables and 5 % for the eight single-output
    variables (one output is noisier than the mean of 60), R2 within 0.02 absolute.  Measured (a development script, since removed):
    aggregate MAE engine 0.01219 vs CPU 0.01210 (0.75 %), single outputs within 2.1 %, and the engine against itself with
    another batch order 0.01219 - the bar is the run-to-run spread of a chaotic optimisation.  The learning-rate drop
    matters: with a constant 1e-3 the final iterate of EITHER implementation moves by 10-35 % on single outputs from one
    run to the next."""
    import copy
    from climsim_amd.data_utils import data_utils
    from oracle.mlp_torch_cpu import TorchMLP
    units, bs, steps = (256, 256), 1024, 1600
    m, cfg, ws = make_model(M, units, bias_scale=0.0, max_batch=4608)
    cpu = TorchMLP(ws, cfg)
    def columns(n, seed):          # synth_columns inputs with a stronger signal in the targets (R2 ~ 0.9 attainable)
        xx, _ = O.synth_columns(n, seed=seed)
        A = np.random.default_rng(7).normal(0, 1 / np.sqrt(124), (124, 128)).astype(np.float32)
        yy = np.tanh(3.0 * xx @ A) * 0.3 + np.random.default_rng(seed + 1000).normal(0, 0.01, (n, 128)).astype(np.float32)
        yy[:, 120:] = np.maximum(yy[:, 120:], 0)
        yy[:, 60:72] = 0
        return xx, yy.astype(np.float32)
    x, y = columns(32 * bs, 31)
    xs, ys = columns(12 * 384, 32)                                  # 12 "timesteps" of the 384-column grid
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    for it in range(steps):
        lo = (it % 32) * bs
        lr = 1e-3 if it < steps * 3 // 4 else 1e-4
        m.train_on_batch(xd[lo:lo + bs], yd[lo:lo + bs], lr)
        _, g = cpu.loss_and_grads(xt[lo:lo + bs], yt[lo:lo + bs])
        cpu.adam(g, lr)
    p_gpu = m.predict(xs)
    with torch.no_grad():
        p_cpu = cpu.forward(torch.from_numpy(xs)).numpy()
    grid, *sets = lowres_assets
    d = data_utils(copy.copy(grid), *sets)
    d.set_to_v1_vars()
    d.input_scoring, d.target_scoring = xs, ys
    d.set_pressure_grid("scoring")
    d.model_names = ["engine", "cpu"]
    d.preds_scoring = {"engine": p_gpu, "cpu": p_cpu}
    d.reweight_target("scoring")
    d.reweight_preds("scoring")
    d.metrics_names = ["MAE", "RMSE", "R2"]
    with np.errstate(all="ignore"):
        d.create_metrics_df("scoring")
    a, b = d.metrics_var_scoring["engine"], d.metrics_var_scoring["cpu"]
    trained = float(np.mean((p_cpu - ys) ** 2)) < 0.5 * float(np.mean(ys ** 2))
    assert trained, "the CPU restatement did not learn; the comparison would be vacuous"
    for v in d.target_vars:
        for name in ("MAE", "RMSE"):
            ga, gb = float(a.loc[v, name]), float(b.loc[v, name])
            assert abs(ga - gb) <= (2e-2 if d.var_lens[v] > 1 else 5e-2) * abs(gb), (v, name, ga, gb)
        ra, rb = float(a.loc[v, "R2"]), float(b.loc[v, "R2"])
        if np.isfinite(ra) and np.isfinite(rb):
            assert abs(ra - rb) <= 2e-2, (v, ra, rb)
        else:
            assert np.isnan(ra) == np.isnan(rb) or np.isinf(ra) == np.isinf(rb), (v, ra, rb)


@pytest.mark.parametrize("flags", [0, 2, 66])             # wide chain | one GEMM per layer (k_gemm_nt2 | k_gemm_nt)
@pytest.mark.parametrize("n_in,n_lin,n_relu,units,n", [(425, 360, 8, (256, 384), 300), (557, 360, 8, (128,), 129), (124, 60, 4, (128, 128), 77)])
def test_v2_shapes_forward_gradients_and_training(M, n_in, n_lin, n_relu, units, n, flags):
    """Other variable sets (hpo_baseline_v2.py:58-101: 425 -> ... -> 368 -> [360 || 8]; v2 full inputs 557): the
    "upper output" layer and the heads are output_length wide, padded to 384 columns inside the engine.  Same tolerances
    as the v1 tests; the padding must stay exactly zero through Adam steps (checked through the weight round trip)."""
    n_out = n_lin + n_relu
    m = M.MLPEmulator(units=units, activation="leakyrelu", optimizer="Adam", input_length=n_in, output_length_lin=n_lin,
                      output_length_relu=n_relu, max_batch=512, seed=None, flags=flags)
    cfg = O.MLPConfig(n_in=n_in, hidden=tuple(units), n_out_lin=n_lin, n_out_relu=n_relu, act="leakyrelu")
    ws = O.glorot_init(cfg, 5)
    rng = np.random.default_rng(9)
    for i in range(1, len(ws), 2):
        ws[i] = rng.normal(0, 0.05, ws[i].shape).astype(np.float32)
    assert m.count_params() == cfg.n_params() == sum(w.size for w in ws)
    m.set_weights(ws)
    for a, b in zip(m.get_weights(), ws):
        np.testing.assert_array_equal(a, b)
    x, y = O.synth_columns(n, seed=3, n_in=n_in, n_out=n_out)
    got = m.predict(x)
    ref16 = O.forward(ws, x, cfg, bf16=True)
    assert got.shape == (n, n_out)
    assert np.max(np.abs(got - ref16)) <= 2e-3 * np.max(np.abs(ref16))
    assert np.all(got[:, n_lin:] >= 0)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    loss = m.loss_grads(xd, yd).cpu().numpy().astype(np.float64)
    ref_loss, ref_mae, ref_g, _ = O.loss_and_grads(ws, x, y, cfg, bf16=True)
    assert loss[0] / (n_out * n) == pytest.approx(ref_loss, rel=2e-3)
    assert loss[1] / (n_out * n) == pytest.approx(ref_mae, rel=2e-3)
    grads = m.get_gradients(1.0 / (n_out * n))
    for i, (g, r) in enumerate(zip(grads, ref_g)):
        assert g.shape == r.shape and rel(g, r) <= 5e-3, (i, rel(g, r))
    ev = m.evaluate(x, y)
    assert ev["mse"] == pytest.approx(ref_loss, rel=2e-3)
    # training: losses track the oracle, padding stays zero (weights round-trip through the padded layout)
    opt = O.Optimizer("Adam")
    w = ws
    lg, lr_ = [], []
    for it in range(8):
        lg.append(m.train_on_batch(xd, yd, 1e-3).cpu().numpy()[0] / (n_out * n))
        w, l, _ = O.train_step(w, opt, x, y, cfg, 1e-3, bf16=True)
        lr_.append(l)
    np.testing.assert_allclose(lg, lr_, rtol=2e-2)
    assert lg[-1] < lg[0]
    after = m.get_weights()
    m.set_weights(after)
    np.testing.assert_allclose(m.predict(x), O.forward(after, x, cfg, bf16=True), atol=2e-3 * np.max(np.abs(ref16)))
