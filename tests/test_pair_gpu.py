"""Pair chain (csrc/chain2.h, CS_FLAG_COOP, 2,049..8,192 columns): a 64-row tile owned by two workgroups that split every
512-wide stage by output columns and exchange halves behind the next stage's contraction (exchange waves, rotated k order,
LDS flags).  Held to the bf16-emulating oracle with the tolerances of tests/test_mlp_gpu.py - ragged row counts, 128-wide
stages in the middle of the stack (computed by both members), every loss - over several steps (monotonic epoch flags), and to
the one-workgroup-per-tile chain on the same inputs.  tests/r02_trip17.sh runs this file a second time with the
write-through payload forced (CS_COOP_WARM=4), the path a pair split over two XCDs takes."""
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


def make(M, units, act, opt="Adam", max_batch=8192, cooperative=True, seed=3, loss="mse"):
    m = M.MLPEmulator(units=units, activation=act, optimizer=opt, max_batch=max_batch, seed=None, cooperative=cooperative, loss=loss)
    cfg = O.MLPConfig(hidden=tuple(units), act=act)
    ws = O.glorot_init(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for i in range(1, len(ws), 2):
        ws[i] = rng.normal(0, 0.05, ws[i].shape).astype(np.float32)
    m.set_weights(ws)
    return m, cfg, ws


@pytest.mark.parametrize("act,n,units", [("leakyrelu", 8192, (512, 512, 512, 512, 512)),      # cfg-MLP at the bench batch: 128 pairs
                                         ("relu", 4133, (512, 512)),                           # ragged: the last pair's tile is half empty
                                         ("leakyrelu", 2300, (512, 128, 512)),                 # a 128-wide stage between split stages
                                         ("relu", 3000, (128, 512, 512, 512))])                # first layer narrow
def test_pair_loss_and_gradients_match_oracle(M, act, n, units):
    m, cfg, ws = make(M, units, act)
    from climsim_amd import _lib
    x, y = O.synth_columns(n, seed=7)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    loss = m.loss_grads(xd, yd, row_idx=perm).cpu().numpy().astype(np.float64)
    ref_loss, ref_mae, ref_g, _ = O.loss_and_grads(ws, x, y, cfg, bf16=True)
    assert loss[0] / (128 * n) == pytest.approx(ref_loss, rel=2e-3)
    assert loss[1] / (128 * n) == pytest.approx(ref_mae, rel=2e-3)
    for i, (g, r) in enumerate(zip(m.get_gradients(1.0 / (128 * n)), ref_g)):
        assert g.shape == r.shape and rel(g, r) <= 5e-3, (i, rel(g, r))
    m.get_weights()                                                   # reports a timed-out wait
    m.close()


def test_pair_training_tracks_oracle_and_the_plain_chain(M):
    units, n = (512, 512, 512), 4096
    a, cfg, ws = make(M, units, "leakyrelu", cooperative=True)
    b, _, _ = make(M, units, "leakyrelu", cooperative=False)
    x, y = O.synth_columns(n, seed=21)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    opt = O.Optimizer("Adam")
    w = ws
    la, lb, lr_ = [], [], []
    for it in range(12):                                             # 12 epochs of the flags
        la.append(float(a.train_on_batch(xd, yd, 1e-3)[0]) / (128 * n))
        lb.append(float(b.train_on_batch(xd, yd, 1e-3)[0]) / (128 * n))
        w, l, _ = O.train_step(w, opt, x, y, cfg, 1e-3, bf16=True)
        lr_.append(l)
    np.testing.assert_allclose(la, lr_, rtol=2e-2)
    np.testing.assert_allclose(la, lb, rtol=5e-3)                   # two decompositions of the same arithmetic
    for wa, wo, w0 in zip(a.get_weights(), w, ws):
        assert rel(wa - w0, wo - w0) <= 2e-2
    # the same handle at another batch size: another pair count (flags start over), then a small batch (coop.h takes over)
    for n2, seed in ((8192, 22), (2560, 23), (1024, 24)):
        x2, y2 = O.synth_columns(n2, seed=seed)
        l2 = a.loss_grads(torch.from_numpy(x2).cuda(), torch.from_numpy(y2).cuda()).cpu().numpy()
        ref2, _, g2, _ = O.loss_and_grads(a.get_weights(), x2, y2, cfg, bf16=True)
        assert l2[0] / (128 * n2) == pytest.approx(ref2, rel=2e-3)
        for g, r in zip(a.get_gradients(1.0 / (128 * n2)), g2):
            assert rel(g, r) <= 5e-3


@pytest.mark.parametrize("loss", ["mae", "huber"])
def test_pair_other_losses_and_predictions(M, loss):
    units, n = (512, 512), 2304
    m, cfg, ws = make(M, units, "relu", max_batch=4096, loss=loss)
    p, _, _ = make(M, units, "relu", max_batch=4096, cooperative=False, loss=loss)
    x, y = O.synth_columns(n, seed=9)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for _ in range(3):
        sa = m.train_on_batch(xd, yd, 1e-3).cpu().numpy()
        sb = p.train_on_batch(xd, yd, 1e-3).cpu().numpy()
        np.testing.assert_allclose(sa, sb, rtol=2e-3)
    np.testing.assert_allclose(m.predict(x), p.predict(x), rtol=0, atol=2e-3)
    for wa, wb, w0 in zip(m.get_weights(), p.get_weights(), ws):
        assert rel(wa - w0, wb - w0) <= 2e-2
