"""Seeded inputs shared by tests/golden/make_online_mlp_golden.py (which feeds them to the REFERENCE model) and
tests/test_online_mlp_*.py (which feed them to the oracle and the HIP engine).  `numpy.random.RandomState` streams are
frozen by NumPy's compatibility policy, so the fixture only has to store the reference's OUTPUTS."""
import numpy as np

CASES = {
    # name: (in_dims, out_dims, hidden_dims, output_prune, strato_lev_out, loss, batch)
    "v2rh_mse_prune12": (557, 368, [128], True, 12, "mse", 32),               # the real v2_rh input width
    "huber_prune15": (64, 368, [128, 256, 128], True, 15, "huber", 32),
    "mae_noprune": (64, 368, [128], False, 15, "mae", 32),
}
LR = 1e-3
N_STEPS = 5


def init_state(name):
    """torch `state_dict` layout of the reference MLP (mlp.py:41-52): linears.{i}.0.weight (out,in) / .bias, final_linear.*"""
    n_in, n_out, hidden, *_ = CASES[name]
    rs = np.random.RandomState(sum(map(ord, name)))
    dims = [n_in, *hidden, n_out]
    sd = {}
    for i in range(len(dims) - 1):
        key = f"linears.{i}.0" if i < len(hidden) else "final_linear"
        sd[key + ".weight"] = (rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
        sd[key + ".bias"] = (rs.standard_normal(dims[i + 1]) * 0.05).astype(np.float32)
    return sd


def batches(name):
    n_in, n_out, _, _, _, loss, nb = CASES[name]
    rs = np.random.RandomState(1000 + sum(map(ord, name)))
    scale_y = 1.5 if loss == "huber" else 0.3          # huber: errors on both sides of 1 (both SmoothL1 branches)
    out = []
    for _ in range(N_STEPS):
        x = (rs.standard_normal((nb, n_in)) * 0.5).astype(np.float32)
        y = (rs.standard_normal((nb, n_out)) * scale_y).astype(np.float32)
        y[:, -8:] = np.abs(y[:, -8:])
        out.append((x, y))
    return out


# ---- the BENCHMARKED topologies on the reference's torch MLP (round 6) ------------------------------------------------
# `MLP(in_dims=124, out_dims=128, hidden_dims=[512]*5+[128], layers=6)` (mlp.py:28-67: Linear -> ReLU per hidden layer, a final
# Linear, ReLU on the last 8 outputs) IS the cfg-MLP of step2_retrain.py:95-126 with activation 'relu' (5 x Dense+act,
# Dense(128)+act, [Dense(120) || Dense(8, relu)] = one 128x128 layer with ReLU on columns 120..127), and
# [768, 640, 512, 640, 640, 128] is the published model (step1_results.csv:170).  These cases run at the batch sizes the bench
# line quotes, i.e. on k_chain_fb<32> + k_wgrad3 + k_optimizer (8192 columns) and on k_chainw_fb (3072).
HOT_CASES = {
    # name: (in_dims, out_dims, hidden_dims, loss, batch)
    "cfg_mlp_b8192": (124, 128, [512] * 5 + [128], "mse", 8192),
    "pub_mlp_b3072": (124, 128, [768, 640, 512, 640, 640, 128], "mse", 3072),
}
HOT_PRED_ROWS = 256          # predictions of this many strided rows are stored
HOT_CORNER = 16              # a 16 x 16 corner and a 16 x 16 strided sample of every weight gradient are stored


def hot_init_state(name):
    """He-scaled kernels (activations keep their size through six ReLU layers) and non-zero biases, torch layout."""
    n_in, n_out, hidden, *_ = HOT_CASES[name]
    rs = np.random.RandomState(sum(map(ord, name)))
    dims = [n_in, *hidden, n_out]
    sd = {}
    for i in range(len(dims) - 1):
        key = f"linears.{i}.0" if i < len(hidden) else "final_linear"
        sd[key + ".weight"] = (rs.standard_normal((dims[i + 1], dims[i])) * np.sqrt(2.0 / dims[i])).astype(np.float32)
        sd[key + ".bias"] = (rs.standard_normal(dims[i + 1]) * 0.05).astype(np.float32)
    return sd


def hot_batches(name):
    """Low-res-shaped columns: 120 profile features ~ N(0, 0.15^2) clipped to [-1, 1], 4 scalars ~ U(-0.5, 0.5) with
    SOLIN zeroed on half of the rows; targets tanh(xA) * 0.3 + noise, the 8 scalar targets >= 0, columns 60..71 exactly zero."""
    n_in, n_out, _, _, nb = HOT_CASES[name]
    rs = np.random.RandomState(2000 + sum(map(ord, name)))
    a = (rs.standard_normal((n_in, n_out)) / np.sqrt(n_in)).astype(np.float32)
    out = []
    for _ in range(N_STEPS):
        x = np.empty((nb, n_in), np.float32)
        x[:, :120] = np.clip(rs.standard_normal((nb, 120)) * 0.15, -1, 1)
        x[:, 120:] = rs.uniform(-0.5, 0.5, (nb, n_in - 120))
        x[rs.random_sample(nb) < 0.5, 121] = 0
        y = (np.tanh(x @ a) * 0.3 + rs.standard_normal((nb, n_out)) * 0.01).astype(np.float32)
        y[:, -8:] = np.maximum(y[:, -8:], 0)
        y[:, 60:72] = 0
        out.append((x, y))
    return out


def hot_pred_rows(name):
    nb = HOT_CASES[name][4]
    return np.arange(0, nb, nb // HOT_PRED_ROWS)[:HOT_PRED_ROWS]


def hot_projections(name, key, shape):
    """Two seeded directions a weight-shaped tensor is projected on (the fixture stores <T, R0>, <T, R1> in float64)."""
    rs = np.random.RandomState(3000 + sum(map(ord, name + key)) % 100000)
    return rs.standard_normal((2, *shape))


def hot_summary(name, key, t):
    """What the fixture keeps of a weight-shaped tensor `t` (torch layout (out, in)): Frobenius norm, two projections, a corner
    and a strided sample - the same function summarises the reference's tensors (make_online_mlp_golden.py) and the tested ones."""
    t64 = np.asarray(t, np.float64)
    r = hot_projections(name, key, t64.shape)
    so, si = max(t64.shape[0] // HOT_CORNER, 1), max(t64.shape[1] // HOT_CORNER, 1)
    return {"fro": np.float64(np.linalg.norm(t64)), "proj": np.array([(t64 * r[0]).sum(), (t64 * r[1]).sum()]),
            "proj_scale": np.float64(np.linalg.norm(t64)),            # |<T,R>| ~ ||T|| for a unit-variance direction
            "corner": np.asarray(t, np.float32)[:HOT_CORNER, :HOT_CORNER].copy(),
            "sample": np.asarray(t, np.float32)[::so, ::si][:HOT_CORNER, :HOT_CORNER].copy()}
