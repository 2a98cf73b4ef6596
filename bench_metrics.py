#!/usr/bin/env python3
"""Device metrics throughput at the size of the low-res scoring split (4,380 time steps x 384 columns, 128 outputs):
predictions/targets resident in HBM -> per-column MAE/RMSE/R2/bias (cs_metrics_columns).  HBM-bound; one JSON line."""
import json
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.metrics import GpuMetrics  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4380
tv = ["ptend_t", "ptend_q0001"] + [f"s{i}" for i in range(8)]
lens = {v: 60 if v.startswith("ptend") else 1 for v in tv}
rng = np.random.default_rng(0)
val = lambda a: types.SimpleNamespace(values=np.asarray(a))  # noqa: E731
du = types.SimpleNamespace(full_vars=False, grid_info={"P0": val(1e5), "hyai": val(np.linspace(0, 0.1, 61)), "hybi": val(np.linspace(0, 1, 61))},
                           target_vars=tv, var_lens=lens, target_energy_conv={v: 1004.0 if v.startswith("ptend") else 1.0 for v in tv},
                           output_scale={v: val(rng.uniform(0.5, 2, lens[v])) for v in tv}, normalize=True, grav=9.8,
                           area_wgt=rng.uniform(0.5, 1.5, 384), num_latlon=384, ps_index=120,
                           input_max={"state_ps": val(1.05e5)}, input_min={"state_ps": val(5e4)}, input_mean={"state_ps": val(9e4)})
gm = GpuMetrics(du)
n = T * 384
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.rand((n, 128), device="cuda", generator=g)
t = torch.rand((n, 128), device="cuda", generator=g)
x = torch.rand((n, 124), device="cuda", generator=g) - 0.5
for _ in range(3):
    gm.column_stats(p, t, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gm.column_stats(p, t, x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
gbs = n * 128 * 8 / dt / 1e9
print(json.dumps({"metric": "metrics rows/sec", "value": round(n / dt, 1), "unit": "rows/s", "rows": n, "ms_per_call": round(dt * 1e3, 3),
                  "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000, 3),
                               "traffic": None}, "note": "includes the surface-pressure gather from the input rows"}))
