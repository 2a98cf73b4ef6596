#!/usr/bin/env python3
"""Many trials per GPU: aggregate training columns/s of K independent cfg-MLP trials (batch 3072, the reference's
largest HPO batch) stepped round-robin on K streams of one MI355X, against one trial alone.  One JSON line."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.hpo import TrialPool  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
g = torch.Generator(device="cuda").manual_seed(0)
n = 64 * B
x = (torch.rand((n, 124), device="cuda", generator=g) - 0.5).contiguous()
y = (torch.rand((n, 128), device="cuda", generator=g) * 0.1).contiguous()
res = {}
for K in (1, 2, 5, 8, 16):
    pool = TrialPool([dict(units=(512,) * 5, activation="leakyrelu", optimizer="Adam", batch_size=B)] * K)
    pool.fit(x, y, epochs=1, steps_per_epoch=10)                      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.fit(x, y, epochs=1, steps_per_epoch=64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[K] = round(K * 64 * B / dt, 1)
    pool.close()
# the same with 8 architectures drawn from the reference's search space (widths 128..1024, 2..12 layers; fixed seed)
import numpy as np  # noqa: E402
from climsim_amd.hpo import sample_trial  # noqa: E402
rng = np.random.default_rng(7)
mix = []
while len(mix) < 8:
    t = sample_trial(rng)
    t["batch_size"] = B
    mix.append(t)
seq = 0.0
for t in mix:                                                         # one after the other
    pool = TrialPool([t])
    pool.fit(x, y, epochs=1, steps_per_epoch=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.fit(x, y, epochs=1, steps_per_epoch=64)
    torch.cuda.synchronize()
    seq += time.perf_counter() - t0
    pool.close()
pool = TrialPool(mix)
pool.fit(x, y, epochs=1, steps_per_epoch=10)
torch.cuda.synchronize()
t0 = time.perf_counter()
pool.fit(x, y, epochs=1, steps_per_epoch=64)
torch.cuda.synchronize()
conc = time.perf_counter() - t0
pool.close()
res_mix = {"trials": [{"units": list(t["units"]), "activation": t["activation"], "optimizer": t["optimizer"]} for t in mix],
           "sequential_columns_per_s": round(8 * 64 * B / seq, 1), "concurrent_columns_per_s": round(8 * 64 * B / conc, 1),
           "speedup": round(seq / conc, 2)}
print(json.dumps({"metric": "aggregate training columns/sec of K concurrent trials", "search_space_mix": res_mix, "unit": "columns/s", "batch": B,
                  "columns_per_s_by_K": res, "speedup_vs_one": {k: round(v / res[1], 2) for k, v in res.items()}}))
