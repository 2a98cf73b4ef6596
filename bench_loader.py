#!/usr/bin/env python3
"""Device loader throughput (SURVEY section 8 config 5 shape: high-res timesteps of 21,600 columns): raw float64 fields
resident in HBM -> normalised float32 rows (cs_loader_stack).  HBM-bound; prints one JSON line.
usage: bench_loader.py [timesteps] [ncol]"""
import json
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.loader import GpuColumnLoader  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NCOL = int(sys.argv[2]) if len(sys.argv) > 2 else 21600
names_in = ["state_t", "state_q0001", "state_ps", "pbuf_SOLIN", "pbuf_LHFLX", "pbuf_SHFLX"]
names_out = ["ptend_t", "ptend_q0001", "cam_out_NETSW", "cam_out_FLWDS", "cam_out_PRECSC", "cam_out_PRECC", "cam_out_SOLS",
             "cam_out_SOLL", "cam_out_SOLSD", "cam_out_SOLLD"]
lens = {v: 60 if v in ("state_t", "state_q0001", "ptend_t", "ptend_q0001") else 1 for v in names_in + names_out}
rng = np.random.default_rng(0)
du = types.SimpleNamespace(input_vars=names_in, target_vars=names_out, var_lens=lens, normalize=True, input_abbrev="mli",
                           output_abbrev="mlo",
                           save_norm=lambda: (rng.normal(0, 1, 124), rng.uniform(0.5, 2, 124), rng.uniform(0.5, 2, 128)))
ld = GpuColumnLoader(du)
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.rand((T, 124, NCOL), device="cuda", dtype=torch.float64, generator=g)
b = torch.rand((T, 128, NCOL), device="cuda", dtype=torch.float64, generator=g)
for _ in range(3):
    ld.stack_raw(a, b)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    x, y = ld.stack_raw(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
cols = T * NCOL
bytes_alg = cols * (124 * 8 + 128 * 8 + 252 * 4)          # 3,024 B per column with f64 sources
print(json.dumps({"metric": "loader columns/sec", "value": round(cols / dt, 1), "unit": "columns/s", "timesteps": T, "ncol": NCOL,
                  "ms_per_call": round(dt * 1e3, 3),
                  "roofline": {"bound": "hbm", "achieved": round(bytes_alg / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(bytes_alg / dt / 8e12, 3), "traffic": None},
                  "note": "includes torch.empty of the outputs; float64 sources, float64 arithmetic, float32 rows"}))
