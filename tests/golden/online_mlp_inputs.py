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
