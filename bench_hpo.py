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
print(json.dumps({"metric": "aggregate training columns/sec of K concurrent trials", "unit": "columns/s", "batch": B,
                  "columns_per_s_by_K": res, "speedup_vs_one": {k: round(v / res[1], 2) for k, v in res.items()}}))
