"""Grouped launches (cs_mlp_group_*, climsim_amd/group.py): K members stepped by ONE launch per kernel kind.  Every member
is held to the ORACLE (oracle/mlp_oracle.py: bf16-emulating forward/backward, Keras / tfa optimiser rules) - not to a
solo run of the engine - with the tolerances of tests/test_mlp_gpu.py: loss 2e-3, loss curves 2e-2 over several steps,
weights after the steps in aggregate.  Covered: members of different architectures, activations (LeakyReLU slopes, ReLU),
optimisers, learning rates, batch sizes and data; members sitting out steps; the wide-chain family (the published
768-640-512-640-640 shape, RPN's ensemble shape); the ELU family; both tile heights of the tuned chain."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import mlp_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd import group, mlp
    return group, mlp


# Round 3: bars set from measurements (tests/conftest.py record_margin -> profiles/r03_test_margins.json; round 2 left 5e-2 /
# 2e-2 here).  Measured maxima over all cases: loss curves 3.2e-4 (tuned), 1.3e-4 (wide), 5.6e-5 (ELU); weight movement
# 1.28e-2 (tuned), 1.3e-3 (ELU), 3.75e-2 (wide: RAdam at lr 2.5e-4 moves the 124 x 768 kernel by ~1e-3 of its norm in 4 steps,
# so bf16 rounding of the gradients is a large share of the movement).
MOVE_TOL = 2e-2
MOVE_TOL_WIDE = 5e-2
CURVE_TOL = 2e-3
from conftest import record_margin  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / (np.linalg.norm(b) + 1e-30))


def make_member(mlp, units, act, opt, seed, max_batch=4096):
    m = mlp.MLPEmulator(units=units, activation=act, optimizer=opt, max_batch=max_batch, seed=None)
    cfg = O.MLPConfig(hidden=tuple(units), act=act)
    ws = O.glorot_init(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for i in range(1, len(ws), 2):
        ws[i] = rng.normal(0, 0.05, ws[i].shape).astype(np.float32)
    m.set_weights(ws)
    return m, cfg, ws


def run_case(G, specs, steps, skip=None, tag="tuned"):
    """specs: [(units, act, opt, lr, n, data_seed)].  Runs `steps` grouped steps (member i sits out step s when
    (i, s) in skip) and the same steps per member through the oracle; returns nothing, asserts."""
    group, mlp = G
    skip = skip or set()
    members, cfgs, wref, opts, data = [], [], [], [], []
    for i, (units, act, opt, lr, n, dseed) in enumerate(specs):
        m, cfg, ws = make_member(mlp, units, act, opt, seed=3 + i)
        members.append(m)
        cfgs.append(cfg)
        wref.append(ws)
        opts.append(O.Optimizer(opt))
        x, y = O.synth_columns(n, seed=dseed)
        data.append((x, y, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))
    g = group.MLPGroup(members)
    lrs = [s[3] for s in specs]
    curves_g = [[] for _ in specs]
    curves_r = [[] for _ in specs]
    for s in range(steps):
        active = [(i, s) not in skip for i in range(len(specs))]
        loss = g.train_on_batch([d[2] for d in data], [d[3] for d in data], lrs, active=active).cpu().numpy().astype(np.float64)
        for i, (units, act, opt, lr, n, dseed) in enumerate(specs):
            if not active[i]:
                continue
            wref[i], l, _ = O.train_step(wref[i], opts[i], data[i][0], data[i][1], cfgs[i], lr, bf16=True)
            curves_g[i].append(loss[i, 0] / (128 * n))
            curves_r[i].append(l)
    for i, m in enumerate(members):
        assert m.iterations == len(curves_r[i])
        record_margin(f"group_{tag}_loss_curve_rel", float(np.max(np.abs(np.asarray(curves_g[i]) / np.asarray(curves_r[i]) - 1.0))))
        np.testing.assert_allclose(curves_g[i], curves_r[i], rtol=CURVE_TOL, err_msg=f"member {i}")
        assert curves_g[i][0] == pytest.approx(curves_r[i][0], rel=2e-3)           # first step: same weights on both sides
        w0 = O.glorot_init(cfgs[i], 3 + i)
        for a, b, z in zip(m.get_weights(), wref[i], w0):
            if z.ndim == 2:                                                            # (biases were re-drawn: compare kernels' movement)
                record_margin(f"group_{tag}_weight_movement_rel", rel(a - z, b - z))
                # Round 6: a kernel's movement is held to the bar where float32 RESOLVES it.  The published widths under RAdam at lr
                # 2.5e-4 move their first kernel by 9e-9 of its norm in four steps (tools/r06_group_wide_diag.py) - a fraction of one
                # ulp per entry: fl(w - d) is then a threshold function of d, and whether an entry moves at all is decided by the last
                # bit of the gradient sum (measured 0.037 against 0.058 for two summation orders of the SAME arithmetic, with
                # first-step gradients at 3e-3 / 4e-3 of the oracle's).  What such a step computes is in the optimiser's moments,
                # which are checked below for every member; the movement is asserted from 16 ulps (1e-6 of the norm) on.
                resolved = np.linalg.norm(b - z) >= 1e-6 * np.linalg.norm(z)
                if resolved:
                    record_margin(f"group_{tag}_weight_movement_resolved_rel", rel(a - z, b - z))
                    assert rel(a - z, b - z) <= (MOVE_TOL_WIDE if tag == "wide" else MOVE_TOL), (i, rel(a - z, b - z))
        if specs[i][2] in ("Adam", "RAdam", "RMSprop"):                                # the optimiser's state after the steps: linear (m) and
            ms, vs, it = m.get_optimizer_state()                                       # quadratic (v) in the gradients of every step taken
            assert it == len(curves_r[i])
            for k, z in enumerate(w0):
                if z.ndim != 2:
                    continue
                if specs[i][2] != "RMSprop":
                    record_margin(f"group_{tag}_first_moment_rel", rel(ms[k], opts[i].m[k]))
                    assert rel(ms[k], opts[i].m[k]) <= 5e-2, (i, k, rel(ms[k], opts[i].m[k]))      # measured 1.9e-2 (tuned, wide), 9e-4 (ELU): profiles/r06_test_margins.json
                record_margin(f"group_{tag}_second_moment_rel", rel(vs[k], opts[i].v[k]))
                assert rel(vs[k], opts[i].v[k]) <= 2e-2, (i, k, rel(vs[k], opts[i].v[k]))          # measured 3.5e-3
    g.close()
    for m in members:
        m.close()


def test_group_members_follow_the_oracle_tuned_chain(G):
    # different depth / width / activation / optimiser / lr / batch / data per member; 5 members x <= 1152 rows -> 32-row tiles
    specs = [((512, 512), "leakyrelu", "Adam", 1e-3, 1024, 11),
             ((128, 256, 512), "relu", "RAdam", 2e-3, 768, 12),
             ((256,), "leakyrelu", "RMSprop", 5e-4, 1152, 13),
             ((512, 128, 128, 256), "relu", "SGD", 1e-2, 300, 14),
             ((512, 512, 512, 512, 512), "leakyrelu", "Adam", 1e-3, 1024, 15)]      # the cfg-MLP
    run_case(G, specs, steps=7)                                                        # crosses RAdam's switch at t = 6


def test_group_members_may_sit_out_steps(G):
    specs = [((256, 256), "relu", "Adam", 1e-3, 512, 21), ((128, 128), "leakyrelu", "Adam", 1e-3, 640, 22),
             ((512,), "relu", "RAdam", 1e-3, 256, 23)]
    run_case(G, specs, steps=5, skip={(0, 1), (2, 0), (2, 3), (1, 4)})


def test_group_elu_family_and_wide_family(G):
    run_case(G, [((256, 128), "elu", "Adam", 1e-3, 500, 31), ((512, 512), "elu", "RMSprop", 1e-3, 384, 32)], steps=4, tag="elu")
    # wide chain: the published lot-147/trial_0027 widths next to other search-space shapes (hpo_baseline_v1.py:66-74)
    run_case(G, [((768, 640, 512, 640, 640), "leakyrelu", "RAdam", 2.5e-4, 768, 41), ((1024, 896), "relu", "Adam", 1e-3, 200, 42),
                 ((384, 384, 384), "leakyrelu", "SGD", 1e-2, 1000, 43)], steps=4, tag="wide")


def test_group_large_total_uses_taller_tiles(G):
    # 4 members x 4096 rows = 16384 rows in one launch -> 64-row tiles (chain_bm's table); per-member parity is unchanged
    specs = [((512, 512, 512), "leakyrelu", "Adam", 1e-3, 4096, 51 + i) for i in range(4)]
    run_case(G, specs, steps=3)


def test_group_equals_solo_engine_run_and_rejects_mixed_families(G):
    group, mlp = G
    x, y = O.synth_columns(1024, seed=5)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    idx = torch.randperm(1024, device="cuda")[:768]
    a, _, _ = make_member(mlp, (512, 256), "leakyrelu", "Adam", 7)
    b, _, _ = make_member(mlp, (512, 256), "leakyrelu", "Adam", 7)
    c, _, _ = make_member(mlp, (128,), "relu", "SGD", 8)
    g = group.MLPGroup([a, c])
    for _ in range(4):
        lg = g.train_on_batch(xd, yd, [1e-3, 1e-2], row_idx=[idx, None]).cpu().numpy()
        ls = b.train_on_batch(xd, yd, 1e-3, row_idx=idx).cpu().numpy()
        np.testing.assert_allclose(lg[0], ls, rtol=1e-4)                                # float atomics order only
    for wa, wb in zip(a.get_weights(), b.get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=2e-4 * max(1.0, float(np.abs(wb).max())))
    # a member of a group still works on its own (checkpoints, evaluation, a solo step)
    ev = a.evaluate(x, y)
    assert np.isfinite(ev["loss"])
    a.train_on_batch(xd, yd, 1e-3)
    g.close()
    from climsim_amd import _lib
    wide, _, _ = make_member(mlp, (640, 384), "relu", "Adam", 9)
    elu, _, _ = make_member(mlp, (128, 128), "elu", "Adam", 9)
    per_layer = mlp.MLPEmulator(units=(128, 128), max_batch=256, seed=1, flags=_lib.CS_FLAG_NO_CHAIN)
    for bad in ([a, wide], [a, elu], [a, per_layer], [a, a]):
        with pytest.raises(_lib.EngineError):
            group.MLPGroup(bad)
    assert group.group_by_family([a, wide, c, elu, per_layer]) == [[4], [0, 2], [1], [3]]


def test_group_evaluation_is_one_launch_and_matches_solo_and_oracle(G):
    """cs_mlp_group_forward (round 3): the validation pass of a trial group as ONE launch per batch.  Every member against
    (a) its own solo evaluate / predict - same kernel bodies, so losses to float-atomics order and predictions bit-identical -
    and (b) the bf16-emulating oracle forward (2e-3 of max |ref|, as tests/test_mlp_gpu.py).  Ragged row counts, members
    sitting out, several batches accumulated."""
    group, mlp = G
    specs = [((512, 512), "leakyrelu"), ((128, 256, 512), "relu"), ((256,), "leakyrelu")]
    members, cfgs, wss = [], [], []
    for i, (units, act) in enumerate(specs):
        m, cfg, ws = make_member(mlp, units, act, "Adam", seed=61 + i, max_batch=1024)
        members.append(m); cfgs.append(cfg); wss.append(ws)
    g = group.MLPGroup(members)
    x, y = O.synth_columns(2500, seed=71)                            # 3 batches of 1024: the last one ragged (452 rows)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    evs = g.evaluate(xd, yd)
    for m, cfg, ws, ev in zip(members, cfgs, wss, evs):
        solo = m.evaluate(x, y)
        assert ev["mse"] == pytest.approx(solo["mse"], rel=1e-5) and ev["mae"] == pytest.approx(solo["mae"], rel=1e-5)
        ref = O.forward(ws, x, cfg, bf16=True)
        assert ev["mse"] == pytest.approx(float(np.mean((ref - y) ** 2)), rel=2e-3)
    # predictions: a (k, n, 128) tensor, member 1 sits out and its slice must stay untouched
    n = 700
    yh = torch.full((3, n, 128), -7.0, device="cuda")
    g.forward_batch(xd[:n], yhat=yh, active=[True, False, True])
    assert float(yh[1].min()) == -7.0 and float(yh[1].max()) == -7.0
    for i in (0, 2):
        np.testing.assert_array_equal(yh[i].cpu().numpy(), members[i].predict(x[:n]))
        ref = O.forward(wss[i], x[:n], cfgs[i], bf16=True)
        assert np.max(np.abs(yh[i].cpu().numpy() - ref)) <= 2e-3 * np.max(np.abs(ref))
    g.close()
    for m in members:
        m.close()


def test_group_follows_head_options_and_refuses_a_member_that_left_its_family(G):
    """a group snapshots its members' chain arguments.  cs_mlp_set_head_options on a member AFTER the group was
    made must reach the grouped steps (config generation counter), and cs_mlp_set_dropout - which moves a tuned-chain model
    to the wide chain - must fail the next grouped call instead of silently training without dropout."""
    group, mlp = G
    from climsim_amd import _lib
    a, cfg_a, ws_a = make_member(mlp, (256, 256), "relu", "SGD", 81, max_batch=512)
    b, _, _ = make_member(mlp, (128,), "relu", "SGD", 82, max_batch=512)
    solo, _, _ = make_member(mlp, (256, 256), "relu", "SGD", 81, max_batch=512)
    g = group.MLPGroup([a, b])
    x, y = O.synth_columns(512, seed=83)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    keep = np.ones(128, np.float32)
    keep[60:75] = 0                                                  # output pruning + mean-absolute-error loss, set AFTER grouping
    a.set_head_options("mae", keep)
    solo.set_head_options("mae", keep)
    lg = g.train_on_batch(xd, yd, [1e-2, 1e-2]).cpu().numpy()
    ls = solo.train_on_batch(xd, yd, 1e-2).cpu().numpy()
    np.testing.assert_allclose(lg[0], ls, rtol=1e-4)                # the group used the member's NEW loss and pruning
    for wa, wb in zip(a.get_weights(), solo.get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=2e-4 * max(1.0, float(np.abs(wb).max())))
    pg = torch.zeros((2, 512, 128), device="cuda")
    g.forward_batch(xd, yhat=pg)
    assert float(pg[0][:, 60:75].abs().max()) == 0.0                # pruned outputs read zero in the grouped prediction too
    a.set_dropout(0.1, seed=5)                                       # tuned chain -> wide chain: another kernel family
    with pytest.raises(_lib.EngineError, match="family"):
        g.train_on_batch(xd, yd, [1e-2, 1e-2])
    with pytest.raises(_lib.EngineError, match="family"):
        g.forward_batch(xd, y=yd)
    g.close()
    for m in (a, b, solo):
        m.close()
