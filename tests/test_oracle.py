"""The CPU oracle against the known answers the reference holds and against an independent
torch-autograd restatement (parity for the training arithmetic is otherwise unpinned)."""
import numpy as np
import pytest

from oracle import mlp_oracle as O


def test_known_answers_published_model():
    # lot-147/trial_0027: step1_results.csv:170 (1,753,472 params), FLOP_calculation.ipynb nb:231
    pub = O.MLPConfig(hidden=(768, 640, 512, 640, 640))
    assert pub.n_params() == 1_753_472
    assert pub.fwd_flops() == 3_503_488
    assert pub.train_flops() == 10_309_632
    cfg = O.MLPConfig()
    assert cfg.n_params() == 1_196_800
    assert cfg.train_flops() == 7_036_928
    ws = O.glorot_init(cfg, 0)
    assert sum(w.size for w in ws) == cfg.n_params()
    assert [w.shape for w in ws[-4:]] == [(128, 120), (120,), (128, 8), (8,)]


def test_bf16_round():
    a = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 1e-40, np.inf, 0.0], dtype=np.float32)
    r = O.bf16_round(a)
    assert r[0] == 1.0
    assert r[1] == 1.0            # tie (1 + 2^-8) rounds to even
    assert r[2] == 1.0078125      # above the tie rounds up
    assert np.isinf(r[5]) and r[6] == 0
    assert np.all((r.view(np.uint32) & 0xFFFF) == 0)
    import torch
    t = torch.randn(4096, dtype=torch.float32)
    np.testing.assert_array_equal(O.bf16_round(t.numpy()), t.to(torch.bfloat16).to(torch.float32).numpy())


@pytest.mark.parametrize("act", ["relu", "elu", "leakyrelu"])
def test_gradients_match_autograd(act):
    from oracle.mlp_torch_cpu import TorchMLP
    import torch
    cfg = O.MLPConfig(hidden=(128, 256, 128), act=act)
    ws = O.glorot_init(cfg, 3)
    for i in range(1, len(ws), 2):
        ws[i] = np.random.default_rng(i).normal(0, 0.05, ws[i].shape).astype(np.float32)
    x, y = O.synth_columns(256, seed=5)
    loss, mae, grads, yhat = O.loss_and_grads(ws, x, y, cfg)
    tm = TorchMLP(ws, cfg)
    tloss, tgrads = tm.loss_and_grads(torch.from_numpy(x), torch.from_numpy(y))
    assert abs(loss - tloss) <= 1e-6 * abs(tloss)
    fused = O.fuse_heads(grads)
    for (gw, gb), tw, tb in zip(fused, tgrads[0::2], tgrads[1::2]):
        np.testing.assert_allclose(gw, tw.numpy(), rtol=2e-4, atol=1e-8)
        np.testing.assert_allclose(gb, tb.numpy(), rtol=2e-4, atol=1e-8)
    assert np.all(yhat[:, 120:] >= 0)


def test_adam_matches_port_and_differs_from_torch_adam():
    from oracle.mlp_torch_cpu import TorchMLP
    import torch
    cfg = O.MLPConfig(hidden=(128, 128))
    ws = O.glorot_init(cfg, 1)
    x, y = O.synth_columns(512, seed=9)
    opt = O.Optimizer("Adam")
    tm = TorchMLP(ws, cfg)
    w = ws
    for it in range(5):
        w, loss, _ = O.train_step(w, opt, x, y, cfg, 1e-3)
        tloss = tm.train_step(torch.from_numpy(x), torch.from_numpy(y), 1e-3)
        assert abs(loss - tloss) <= 1e-5 * abs(tloss)
    for (a, b), tw, tb in zip(O.fuse_heads(w), tm.params[0::2], tm.params[1::2]):
        np.testing.assert_allclose(a, tw.detach().numpy(), rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(b, tb.detach().numpy(), rtol=1e-3, atol=2e-6)


def test_optimizer_rules_scalar():
    # one scalar parameter, hand-computed first steps
    g = np.array([0.5], dtype=np.float32)
    w0 = [np.array([1.0], dtype=np.float32)]
    adam = O.Optimizer("Adam")
    w = adam.apply(w0, [g], 0.1)
    # t=1: m=0.05, v=2.5e-4, alpha=0.1*sqrt(1-.999)/(1-.9) -> step = alpha*m/(sqrt(v)+eps) ~ 0.1
    assert abs(w[0][0] - (1.0 - 0.1 * 0.5 / (0.5 + 1e-7 / np.sqrt(0.001)))) < 1e-6
    sgd = O.Optimizer("SGD")
    assert abs(sgd.apply(w0, [g], 0.1)[0][0] - 0.95) < 1e-7
    rms = O.Optimizer("RMSprop")
    assert abs(rms.apply(w0, [g], 0.1)[0][0] - (1 - 0.1 * 0.5 / np.sqrt(0.025 + 1e-7))) < 1e-6
    radam = O.Optimizer("RAdam")
    w = radam.apply(w0, [g], 0.1)
    assert abs(w[0][0] - 0.95) < 1e-6          # sma_1 < 5: un-rectified step lr*m_hat = 0.1*0.5
    for _ in range(5):
        w = radam.apply(w, [g], 0.1)
    t = 6
    sma_inf = 2 / (1 - 0.999) - 1
    sma_t = sma_inf - 2 * t * 0.999 ** t / (1 - 0.999 ** t)
    assert sma_t >= 5 and radam.it == 6


def test_cyclical_lr_triangular2():
    s = 16
    assert O.cyclical_lr(0, step_size=s) == pytest.approx(2.5e-4)
    assert O.cyclical_lr(s, step_size=s) == pytest.approx(2.5e-3)
    assert O.cyclical_lr(2 * s, step_size=s) == pytest.approx(2.5e-4)
    assert O.cyclical_lr(3 * s, step_size=s) == pytest.approx(2.5e-4 + (2.5e-3 - 2.5e-4) / 2)
    assert O.cyclical_lr(5 * s, step_size=s) == pytest.approx(2.5e-4 + (2.5e-3 - 2.5e-4) / 4)


def test_normalise_inf_nan_rule():
    x = np.array([[1.0, 2.0, 3.0]], dtype=np.float32)
    out = O.normalise(x, np.array([1, 2, 1], np.float32), np.array([2, 0, 0], np.float32))
    np.testing.assert_array_equal(out, [[0.0, 0.0, 0.0]])   # 0/2, 0/0 -> nan -> 0, 2/0 -> inf -> 0


def test_bf16_mode_close_to_fp32():
    cfg = O.MLPConfig(hidden=(256, 256))
    ws = O.glorot_init(cfg, 2)
    x, y = O.synth_columns(512, seed=4)
    l32, _, g32, y32 = O.loss_and_grads(ws, x, y, cfg)
    l16, _, g16, y16 = O.loss_and_grads(ws, x, y, cfg, bf16=True)
    assert abs(l16 - l32) < 0.02 * l32
    assert np.max(np.abs(y16 - y32)) < 0.02 * np.max(np.abs(y32))


def test_cnn_oracle_known_answers():
    from oracle import cnn_oracle as CO
    shapes = CO.cnn_shapes()
    assert sum(int(np.prod(s)) for s in shapes) == 13_215_420          # BASELINE.md CNN model size
    ws = CO.glorot_cnn(seed=1, depth=2, channels=32)
    x3 = np.random.default_rng(0).normal(0, 0.3, (3, 60, 6)).astype(np.float32)
    y = CO.forward(ws, x3, depth=2)
    assert y.shape == (3, 60, 10) and np.all(y[:, :, 2:] >= 0)
    # 'same' zero padding never mixes columns: predictions of a column do not depend on its neighbours
    y1 = CO.forward(ws, x3[1:2], depth=2)
    np.testing.assert_allclose(y[1:2], y1, rtol=1e-6, atol=1e-7)
    t = np.zeros_like(y)
    assert CO.mae_adjusted(t, y) == pytest.approx(np.abs(y[:, :, :2]).mean() * 120 / 128 + np.abs(y[:, :, 2:]).mean() * 8 / 128)


def test_cnn_oracle_gradients_vs_finite_differences():
    """The CNN training oracle (torch autograd) against central differences of its own forward loss, and the
    dropout hash against its Bernoulli law."""
    from oracle import cnn_oracle as CO
    ws = CO.glorot_cnn(seed=2, bias_scale=0.05, depth=2, channels=16)
    _, _, x3, y3 = CO.synth_cnn_columns(4, seed=1)
    out, g = CO.loss_and_grads(ws, x3, y3, depth=2, loss="mse", rate=0.25, seed=4)
    rng = np.random.default_rng(0)
    for trial in range(3):                       # directional derivatives over all tensors at once
        d = [rng.normal(0, 1, w.shape).astype(np.float32) for w in ws]
        eps = 2e-3
        vals = []
        for sgn in (+1, -1):
            w2 = [a + np.float32(sgn * eps) * b for a, b in zip(ws, d)]
            vals.append(CO.loss_and_grads(w2, x3, y3, depth=2, loss="mse", rate=0.25, seed=4)[0]["mse_adjusted"])
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float(sum((a.astype(np.float64) * b).sum() for a, b in zip(g, d)))
        assert abs(fd - an) <= 3e-2 * abs(an) + 1e-5, (trial, fd, an)
    keep = CO.dropout_keep(123, 3, 600, 406, 0.175)
    assert abs(keep.mean() - 0.825) < 3e-3
    assert not np.array_equal(keep, CO.dropout_keep(124, 3, 600, 406, 0.175))
    # inference forward == training forward at rate 0
    np.testing.assert_allclose(CO.loss_and_grads(ws, x3, y3, depth=2)[0]["pred"], CO.forward(ws, x3, depth=2), atol=1e-6)
    assert abs(out["mae_adjusted"] - CO.mae_adjusted(y3, out["pred"])) < 1e-6


def _torch_opt_run(kind, ws, grads_seq, lr, **kw):
    """The same gradient sequence through torch.optim (an implementation the builder did not write)."""
    import torch
    ps = [torch.nn.Parameter(torch.from_numpy(w.copy())) for w in ws]
    opt = {"RMSprop": torch.optim.RMSprop, "RAdam": torch.optim.RAdam, "SGD": torch.optim.SGD}[kind](ps, lr=lr, **kw)
    for grads in grads_seq:
        for p, g in zip(ps, grads):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
    return [p.detach().numpy() for p in ps]


def test_rmsprop_radam_sgd_against_torch_optim():
    """Independent pin of the optimiser restatements: with epsilon = 0 the Keras-2.11 RMSprop and the
    tfa-0.19 RectifiedAdam rules are algebraically torch.optim.RMSprop(alpha=rho) / torch.optim.RAdam, so the oracle must
    reproduce torch's trajectories; the ONLY documented differences are where epsilon enters, asserted explicitly below:
      RMSprop  Keras:  g * rsqrt(v + eps)                  torch:  g / (sqrt(v) + eps)
      RAdam    tfa:    r*m_hat / (sqrt(v/bc2) + eps)       torch:  r*m_hat*sqrt(bc2) / (sqrt(v) + eps)
                       (= eps_tfa  <->  eps_torch / sqrt(bc2)); threshold sma_t >= 5 (tfa) vs rho_t > 5 (torch)."""
    rng = np.random.default_rng(0)
    ws = [rng.normal(0, 1, (17, 9)).astype(np.float32), rng.normal(0, 1, (9,)).astype(np.float32)]
    seq = [[rng.normal(0, 1e-2, w.shape).astype(np.float32) for w in ws] for _ in range(12)]   # crosses RAdam's switch at t = 6
    lr = 3e-3
    for kind, kw in (("RMSprop", dict(alpha=0.9, eps=0.0)), ("RAdam", dict(betas=(0.9, 0.999), eps=0.0)), ("SGD", {})):
        opt = O.Optimizer(kind, eps=0.0)
        w = [a.copy() for a in ws]
        for grads in seq:
            w = opt.apply(w, grads, lr)
        ref = _torch_opt_run(kind, ws, seq, lr, **kw)
        for a, b, w0 in zip(w, ref, ws):
            np.testing.assert_allclose(a - w0, b - w0, rtol=2e-4, atol=5e-7)       # the accumulated UPDATE agrees (4 float32 ulp of |w| ~ 1)
    # epsilon placement, one scalar step from zero state: v = (1-rho) g^2, Keras eps inside the root, torch outside
    g, eps = np.float32(1e-3), 1e-7
    k = O.Optimizer("RMSprop", eps=eps).apply([np.zeros(1, np.float32)], [np.full(1, g)], 1.0)[0][0]
    t = _torch_opt_run("RMSprop", [np.zeros(1, np.float32)], [[np.full(1, g)]], 1.0, alpha=0.9, eps=eps)[0][0]
    v = 0.1 * float(g) ** 2
    assert k == pytest.approx(-float(g) / np.sqrt(v + eps), rel=1e-6)
    assert t == pytest.approx(-float(g) / (np.sqrt(v) + eps), rel=1e-6)
    assert abs(k - t) > 1e-4 * abs(t)                         # the two conventions are distinguishable at this gradient scale
    # RAdam, rectified regime (t = 8 > 5): tfa's eps corresponds to torch's eps / sqrt(1 - beta2^t)
    g = np.float32(1e-5)                                      # eps = 1 % of sqrt(v_hat) in tfa's form, 11 % in torch's
    seq1 = [[np.full(1, g)] for _ in range(8)]
    w_tfa = [np.zeros(1, np.float32)]
    o = O.Optimizer("RAdam", eps=eps)
    for grads in seq1[:-1]:
        w_tfa = o.apply(w_tfa, grads, 1.0)
    before = w_tfa[0][0]
    step_tfa = o.apply(w_tfa, seq1[-1], 1.0)[0][0] - before
    b2 = float(np.float32(0.999))                            # the variable dtype is float32: beta2 = 0.99899995..., and sma_t
    bc2 = 1 - b2 ** 8                                         # (a difference of two ~2000s) feels its 5e-8
    m_hat, vv = float(g), float(g) ** 2 * bc2                 # constant gradient: m/bc1 = g, v = g^2 * bc2
    sma_inf = 2 / (1 - b2) - 1
    sma_t = sma_inf - 2 * 8 * b2 ** 8 / bc2
    r = np.sqrt((sma_t - 4) / (sma_inf - 4) * (sma_t - 2) / (sma_inf - 2) * sma_inf / sma_t)
    # (float32 evaluation of sma_t moves r by ~4e-4 at t = 8: the bar is 2e-3, the two conventions are 9 % apart)
    assert step_tfa == pytest.approx(-r * m_hat / (np.sqrt(vv / bc2) + eps), rel=2e-3)
    torch_step = -r * m_hat * np.sqrt(bc2) / (np.sqrt(vv) + eps)
    assert abs(step_tfa - torch_step) > 5e-2 * abs(torch_step)   # eps/sqrt(bc2) = 11x larger effective epsilon in torch's form
    t8 = _torch_opt_run("RAdam", [np.zeros(1, np.float32)], seq1[:-1], 1.0, betas=(0.9, 0.999), eps=eps)[0][0]
    t9 = _torch_opt_run("RAdam", [np.zeros(1, np.float32)], seq1, 1.0, betas=(0.9, 0.999), eps=eps)[0][0]
    assert (t9 - t8) == pytest.approx(torch_step, rel=2e-3)      # and torch.optim.RAdam indeed takes the other one


def test_cnn_metric_restatements_against_a_loop():
    """oracle/cnn_oracle.py: continuous_ranked_probability_score (hpo_train.py:83-111) and Keras' categorical accuracy, against
    plain loops over a small tensor and against two closed forms."""
    from oracle import cnn_oracle as CO
    rs = np.random.RandomState(0)
    yt, yp = rs.standard_normal((3, 5, 10)), rs.standard_normal((3, 5, 10))
    tot = 0.0
    for b in range(3):
        for l in range(5):
            a = sum(abs(yp[b, l, j] - yt[b, l, j]) for j in range(10)) / 10
            d = sum(abs(yp[b, l, i] - yp[b, l, j]) for i in range(10) for j in range(10)) / 100
            tot += a - 0.5 * d
    assert abs(CO.continuous_ranked_probability_score(yt, yp) - tot / 15) < 1e-12
    flat = np.full((2, 4, 10), 0.3)                              # a degenerate "ensemble": the score is the mean absolute error
    assert abs(CO.continuous_ranked_probability_score(yt[:2, :4], flat) - np.abs(flat - yt[:2, :4]).mean()) < 1e-12
    acc = np.mean([[np.argmax(yt[b, l]) == np.argmax(yp[b, l]) for l in range(5)] for b in range(3)])
    assert CO.categorical_accuracy(yt, yp) == acc and CO.categorical_accuracy(yt, yt) == 1.0
    tie = np.zeros((1, 1, 10))
    assert CO.categorical_accuracy(tie, yp[:1, :1] * 0 + np.arange(10)[::-1]) == 1.0      # first index on ties: 0 == argmax of a decreasing row


def test_cnn_loss_functions_match_the_reference_function_bodies():
    """oracle/cnn_oracle.py's `mae_adjusted`, `mse_adjusted` and `continuous_ranked_probability_score` against
    tests/golden/cnn_loss_golden.npz - made by tests/golden/make_cnn_loss_golden.py, which executes the SOURCE of the reference's own three
    functions (baseline_models/CNN/training/hpo_train.py:83-121, taken out of the file by ast) with `tf` / `K` bound to numpy namesakes:
    terms, axes, slices and the 120/128, 8/128 weights are the reference's text.  float64, 1e-12."""
    import os
    sys_path = os.path.join(os.path.dirname(__file__), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cnn_loss_golden", os.path.join(sys_path, "make_cnn_loss_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                                    # (only its seeded `inputs()` is used here: /root/reference is not read)
    from oracle import cnn_oracle as CO
    gold = np.load(os.path.join(sys_path, "cnn_loss_golden.npz"))
    for key, (yt, yp) in gen.inputs().items():
        assert abs(CO.mae_adjusted(yt, yp) - float(gold[f"{key}/mae_adjusted"])) <= 1e-12
        assert abs(CO.mse_adjusted(yt, yp) - float(gold[f"{key}/mse_adjusted"])) <= 1e-12
        assert abs(CO.continuous_ranked_probability_score(yt, yp) - float(gold[f"{key}/continuous_ranked_probability_score"])) <= 1e-12
