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


def run_case(G, specs, steps, skip=None):
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
        np.testing.assert_allclose(curves_g[i], curves_r[i], rtol=2e-2, err_msg=f"member {i}")
        assert curves_g[i][0] == pytest.approx(curves_r[i][0], rel=2e-3)           # first step: same weights on both sides
        w0 = O.glorot_init(cfgs[i], 3 + i)
        for a, b, z in zip(m.get_weights(), wref[i], w0):
            if z.ndim == 2:                                                            # (biases were re-drawn: compare kernels' movement)
                assert rel(a - z, b - z) <= 5e-2, (i, rel(a - z, b - z))
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
    run_case(G, [((256, 128), "elu", "Adam", 1e-3, 500, 31), ((512, 512), "elu", "RMSprop", 1e-3, 384, 32)], steps=4)
    # wide chain: the published lot-147/trial_0027 widths next to other search-space shapes (hpo_baseline_v1.py:66-74)
    run_case(G, [((768, 640, 512, 640, 640), "leakyrelu", "RAdam", 2.5e-4, 768, 41), ((1024, 896), "relu", "Adam", 1e-3, 200, 42),
                 ((384, 384, 384), "leakyrelu", "SGD", 1e-2, 1000, 43)], steps=4)


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
