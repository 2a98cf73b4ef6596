#!/usr/bin/env python3
"""CNN prediction throughput (BASELINE.json config 3 shape: depth 12, width 406, 60 levels) on one MI355X.
Secondary measurement - bench.py stays the MLP training metric.  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.cnn import CNNEmulator  # noqa: E402
from climsim_amd.cnn import _shapes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = CNNEmulator(depth=12, channel_width=406, max_batch=B)
rng = np.random.default_rng(0)
m.set_weights([(rng.standard_normal(s) * (0.6 * (2.0 / (np.prod(s[:-1]) + s[-1])) ** 0.5 if len(s) > 1 else 0.0)).astype(np.float32)
               for s in _shapes(12, 406)])
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
for _ in range(3):
    m.predict(x, as_numpy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    m.predict(x, as_numpy=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flops = 1.584e9 * B            # forward FLOP per column, BASELINE.md
print(json.dumps({"metric": "CNN prediction columns/sec", "value": round(B / dt, 1), "unit": "columns/s", "batch": B,
                  "ms_per_batch": round(dt * 1e3, 3), "tflops_algorithmic": round(flops / dt / 1e12, 1),
                  "note": "channels padded 406->448(K)/512(N); per-layer conv-GEMM kernels"}))
