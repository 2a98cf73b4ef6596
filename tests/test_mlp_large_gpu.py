"""Oracle parity at the batch sizes the published throughput numbers are measured on (SURVEY section 8d:
cfg-MLP at B in {1024, 8192, 65536}; the published model at its batch 3072) with DEFAULT engine flags, so that the kernels
the engine picks by batch size are the ones under test:

  cfg-MLP   8192  k_chain_fb<32> + k_wgrad3 (5 row splits)            - the bench.py workload
  cfg-MLP  16384  k_chain_fb<64> + k_wgrad3 with >= 7 row splits
  cfg-MLP  24576  k_chain_fb<128> + k_wgrad2 (256x256 LDS-DMA tiles, n >= 22528)
  cfg-MLP  65536  k_chain_fb<128>, two rounds of workgroups + k_wgrad2
  pub-MLP   3072  k_chainw_fb at the published model's batch (step1_results.csv:170)
  pub-MLP  16384  k_chainw_fb + k_wgrad2-free wide path

The oracle (oracle/mlp_oracle.py, bf16 rounding points emulated, float64 accumulation) takes ~1 minute for 65536 x 5x512 on
8 host cores.  Tolerances are the ones of tests/test_mlp_gpu.py: loss 2e-3 relative, every gradient tensor
||d||/||ref|| <= 5e-3 (accumulation order + rare 1-ulp bf16 flips)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import mlp_oracle as O  # noqa: E402

CFG = (512,) * 5
PUB = (768, 640, 512, 640, 640)


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


@pytest.mark.parametrize("units,n", [(CFG, 8192), (CFG, 16384), (CFG, 24576), (CFG, 65536), (PUB, 3072), (PUB, 16384)])
def test_default_kernels_at_published_batch_sizes_match_oracle(M, units, n):
    m = M.MLPEmulator(units=units, activation="leakyrelu", optimizer="Adam", max_batch=n, seed=None)
    cfg = O.MLPConfig(hidden=tuple(units))
    ws = O.glorot_init(cfg, 3)
    rng = np.random.default_rng(103)
    for i in range(1, len(ws), 2):
        ws[i] = rng.normal(0, 0.05, ws[i].shape).astype(np.float32)
    m.set_weights(ws)
    x, y = O.synth_columns(n, seed=7)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    # the same rows through a permutation (the path fit() and bench.py take: gather inside the first kernel)
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    loss = m.loss_grads(xd, yd, row_idx=perm).cpu().numpy().astype(np.float64)
    got = m.get_gradients(1.0 / (128 * n))
    ref_loss, ref_mae, ref_g, ref_yhat = O.loss_and_grads(ws, x, y, cfg, bf16=True)
    assert loss[0] / (128 * n) == pytest.approx(ref_loss, rel=2e-3)
    assert loss[1] / (128 * n) == pytest.approx(ref_mae, rel=2e-3)
    for i, (g, r) in enumerate(zip(got, ref_g)):
        assert g.shape == r.shape
        assert rel(g, r) <= 5e-3, (i, rel(g, r))
    # prediction through the forward-only chain of the same tile height
    # (the small-batch bar max|d| <= 2e-3 max|ref| is a bound on ONE bf16 flip of a late activation; the maximum over
    #  n x 128 >= 1 M outputs sees rarer, larger flips - measured 3.0e-3 at 8192 rows - so: L2 error <= 1e-3, max <= 6e-3)
    pred = m.predict(xd, as_numpy=False).cpu().numpy()
    assert rel(pred, ref_yhat) <= 1e-3
    assert np.max(np.abs(pred - ref_yhat)) <= 6e-3 * np.max(np.abs(ref_yhat))
    # one optimiser step on top: the update applied to every parameter is the oracle's (Adam normalises, so compare the
    # step direction in aggregate: a near-zero gradient may change sign, so cos >= 0.98) and the loss after the step went down
    opt = O.Optimizer("Adam")
    w1 = opt.apply(ws, ref_g, 1e-3)
    m.train_on_batch(xd, yd, 1e-3, row_idx=perm)
    after = m.get_weights()
    for a, b, w0 in zip(after, w1, ws):
        da, db = (a - w0).ravel().astype(np.float64), (b - w0).ravel().astype(np.float64)
        cos = float(da @ db / (np.linalg.norm(da) * np.linalg.norm(db) + 1e-300))
        assert cos >= 0.98, cos
    m.close()


@pytest.mark.parametrize("n,opt", [(2500, "Adam"), (8192, "Adam"), (8192, "RAdam"), (6000, "SGD")])
def test_training_step_without_gradient_atomics_equals_the_two_call_form(M, n, opt):
    """cs_mlp_train_step lets k_wgrad3 STORE its row splits into separate buffers that the optimiser kernel adds up
    (WgradArgs.plain, 2 or 3 splits); cs_mlp_loss_grads + cs_mlp_apply - what data-parallel training calls - accumulates with
    float atomics into one buffer.  Same gradients up to the order of float additions, so the same weights after several
    steps (every optimiser family reads the extra buffers).  The two-call form hands the gradient buffer back zeroed (its next
    launch adds to it); after a train_step its content is unspecified (round 5: every element is stored again by the next step, so
    the optimiser kernel no longer zeroes it) - what must hold is that a gradient read afterwards sees one clean buffer."""
    cfg = O.MLPConfig(hidden=CFG)
    ws = O.glorot_init(cfg, 5)
    a = M.MLPEmulator(units=CFG, optimizer=opt, max_batch=n, seed=None)
    b = M.MLPEmulator(units=CFG, optimizer=opt, max_batch=n, seed=None)
    a.set_weights(ws); b.set_weights(ws)
    x, y = O.synth_columns(n, seed=11)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for it in range(4):
        perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(it))
        la = a.train_on_batch(xd, yd, 1e-3, row_idx=perm).cpu().numpy()
        lb = b.loss_grads(xd, yd, row_idx=perm).cpu().numpy()
        b.apply_gradients(1e-3, 1.0 / (128 * n))
        np.testing.assert_allclose(la, lb, rtol=2e-3)
    assert float(b.gradient_tensor().abs().max()) == 0.0
    for wa, wb, w0 in zip(a.get_weights(), b.get_weights(), ws):
        assert rel(wa - w0, wb - w0) <= 2e-3, rel(wa - w0, wb - w0)
    # and a gradient read after a training step still sees one clean buffer
    g1 = a.loss_grads(xd, yd).cpu().numpy()
    ga = a.get_gradients(1.0)
    b.set_weights(a.get_weights())
    b.loss_grads(xd, yd)
    for u, v in zip(ga, b.get_gradients(1.0)):
        assert rel(u, v) <= 1e-4


@pytest.mark.parametrize("n", [8192, 4096 + 17, 100, 33])
def test_continuous_run_equals_one_queue_per_stage(M, monkeypatch, n):
    """chain_trunk (csrc/chain.h) carries the weight queue from one 512-wide stage into the next behind waits that COUNT the memory
    operations issued in between (round-4 advisor finding: nothing but the code's shape enforces that count).  CS_CHAIN_TRUNK=0
    runs the same stages with one queue per stage (every wait sized by chain_mma): same arithmetic, same order - loss sums,
    every gradient (to rounding: see below), the weights after three steps and the predictions (bit for bit), also for row counts that leave the last
    32-row tile partly empty."""
    cfg = O.MLPConfig(hidden=CFG)
    ws = O.glorot_init(cfg, 7)
    x, y = O.synth_columns(n, seed=13)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = {}
    for trunk in ("1", "0"):
        monkeypatch.setenv("CS_CHAIN_TRUNK", trunk)
        m = M.MLPEmulator(units=CFG, max_batch=max(n, 128), seed=None)
        m.set_weights(ws)
        sums = m.loss_grads(xd, yd).cpu().numpy()
        g = [a.copy() for a in m.get_gradients(1.0)]
        for _ in range(3):
            m.train_on_batch(xd, yd, 1e-3)
        out[trunk] = (sums, g, [w.copy() for w in m.get_weights()], np.asarray(m.predict(xd)))
        m.close()
    # (the two-call form adds its row splits and the workgroups' loss sums with float atomics: their ORDER is not fixed, so these two
    #  agree to rounding; the one-call steps store their splits and add them in a fixed order: bit for bit)
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-6)
    for a, b in zip(out["1"][1], out["0"][1]):
        assert rel(a, b) <= 1e-6
    for a, b in zip(out["1"][2], out["0"][2]):
        assert np.array_equal(a, b)
    assert np.array_equal(out["1"][3], out["0"][3])


@pytest.mark.parametrize("n", [8192, 1024])
def test_accumulating_gradient_call_after_a_step_starts_clean(M, n):
    """Round-5 advisor finding: a one-call step may leave its (applied) gradients in the flat buffer; `loss_grads(accumulate=True)` as
    the FIRST micro-batch after such a step - or after a grouped step, or on a fresh handle - must not add onto them.  Two
    accumulated micro-batches after a step == the same two on a model that was never stepped but holds the same weights."""
    from climsim_amd.group import MLPGroup
    cfg = O.MLPConfig(hidden=CFG)
    ws = O.glorot_init(cfg, 9)
    x, y = O.synth_columns(n, seed=17)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    h = n // 2

    def two_micro_batches(m):
        m.loss_grads(xd[:h], yd[:h], accumulate=True)
        m.loss_grads(xd[h:], yd[h:], accumulate=True)
        return m.get_gradients(1.0)

    a = M.MLPEmulator(units=CFG, max_batch=n, seed=None)
    a.set_weights(ws)
    a.train_on_batch(xd, yd, 1e-3)
    got = two_micro_batches(a)
    b = M.MLPEmulator(units=CFG, max_batch=n, seed=None)
    b.set_weights(a.get_weights())
    b.set_optimizer_state(*a.get_optimizer_state())
    b.loss_grads(xd[:h], yd[:h])
    b.loss_grads(xd[h:], yd[h:], accumulate=True)
    want = b.get_gradients(1.0)
    for u, v in zip(got, want):
        assert rel(u, v) <= 1e-4, rel(u, v)
    # fresh handle, first call accumulating
    c = M.MLPEmulator(units=CFG, max_batch=n, seed=None)
    c.set_weights(a.get_weights())
    for u, v in zip(two_micro_batches(c), want):
        assert rel(u, v) <= 1e-4, rel(u, v)
    # after a grouped step (one row split stores the gradients and leaves them in place)
    g = MLPGroup([a, c])
    g.train_on_batch(xd[:h], yd[:h], 1e-3)
    b.set_weights(a.get_weights())
    b.loss_grads(xd[:h], yd[:h])
    b.loss_grads(xd[h:], yd[h:], accumulate=True)
    for u, v in zip(two_micro_batches(a), b.get_gradients(1.0)):
        assert rel(u, v) <= 1e-4, rel(u, v)
    g.close()
    for m in (a, b, c):
        m.close()


@pytest.mark.parametrize("units,n,exact", [(CFG, 8192, True), (CFG, 1000, True), (PUB, 3072, True), (CFG, 16384, False)])
def test_wgrad3_asm_loop_equals_the_builtin_loop(M, monkeypatch, units, n, exact):
    """k_wgrad3's contraction is hand-scheduled asm (round 5): no compiler hazard check sees its MFMAs, and it reloads fragment
    registers right behind the MFMAs that read them (round-5 advisor finding).  CS_WGRAD3_ASM=0 runs the same contraction through
    builtins - hipcc's own waits and hazard handling - with the same steps in the same order per accumulator.  Inside a one-call
    step with up to three row splits every split STORES its tile and k_optimizer adds the buffers in a fixed order: the weights
    after three SGD steps agree BIT FOR BIT (8192 rows = the bench's launch: 64-row stages, bias tiles; 1000 = a ragged last stage;
    the published widths at their batch).  16384 rows run 32-row stages, two workgroups per CU, with seven splits added by float
    atomics: equal to the order of those additions."""
    cfg = O.MLPConfig(hidden=tuple(units))
    ws = O.glorot_init(cfg, 21)
    x, y = O.synth_columns(n, seed=23)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CS_WGRAD3_ASM", mode)
        m = M.MLPEmulator(units=units, optimizer="SGD", max_batch=max(n, 128), seed=None)
        m.set_weights(ws)
        for _ in range(3):
            m.train_on_batch(xd, yd, 1e-2)
        out[mode] = [w.copy() for w in m.get_weights()]
        m.close()
    for a, b, w0 in zip(out["1"], out["0"], ws):
        if exact:
            assert np.array_equal(a, b)
        else:
            assert rel(a - w0, b - w0) <= 1e-4                       # (float atomics: the order of seven partial sums; a wrong fragment would show at 1e-2)
        assert not np.array_equal(a, w0)                            # (the steps moved every tensor)
