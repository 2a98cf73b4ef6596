#!/usr/bin/env python3
"""Online-testing MLP (557 -> 9 x 256 -> 368, ReLU, output pruning 12, huber, torch-flavoured Adam: the configuration of
online_testing/baseline_models/MLP_v2rh/training/conf/config_single.yaml) - training columns/s on one MI355X at the
reference's batch 1024 and at 8192, next to the same step in torch on the host cores (the reference's own code path is
torch: nn.Linear stack + nn.SmoothL1Loss + torch.optim.Adam).  One JSON line.  Synthetic v2_rh-shaped columns."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.online_mlp import MLP, output_keep_mask  # noqa: E402

N_IN, N_OUT, HID = 557, 368, [256] * 9
FLOPS = 2 * 3 * (N_IN * 256 + 8 * 256 * 256 + 256 * N_OUT) - 2 * N_IN * 256       # fwd + wgrad + dgrad (no dgrad into x)
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for B in (1024, 8192):
    n = 32 * B
    x = ((torch.rand((n, N_IN), device="cuda", generator=g) - 0.5)).contiguous()
    y = ((torch.rand((n, N_OUT), device="cuda", generator=g) - 0.5) * 2).contiguous()
    m = MLP(N_IN, N_OUT, HID, 9, output_prune=True, strato_lev_out=12, loss="huber", max_batch=B, seed=0)
    eng = m.engine
    for s in range(10):
        eng.train_on_batch(x[(s % 32) * B:(s % 32 + 1) * B], y[(s % 32) * B:(s % 32 + 1) * B], 1e-4)
    torch.cuda.synchronize()
    steps = 100
    t0 = time.perf_counter()
    for s in range(steps):
        eng.train_on_batch(x[(s % 32) * B:(s % 32 + 1) * B], y[(s % 32) * B:(s % 32 + 1) * B], 1e-4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[B] = {"columns_per_s": round(steps * B / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
              "tflops_algorithmic": round(FLOPS * steps * B / dt / 1e12, 1)}
    eng.close()

# host-core baseline: the reference step restated with torch modules (float32, all threads), batch 1024
keep = torch.from_numpy(output_keep_mask(N_OUT, True, 12))
layers = []
d = N_IN
for h in HID:
    layers += [torch.nn.Linear(d, h), torch.nn.ReLU()]
    d = h
trunk, final = torch.nn.Sequential(*layers), torch.nn.Linear(d, N_OUT)
params = list(trunk.parameters()) + list(final.parameters())
opt = torch.optim.Adam(params, lr=1e-4)
crit = torch.nn.SmoothL1Loss()
xc, yc = torch.rand(1024, N_IN) - 0.5, (torch.rand(1024, N_OUT) - 0.5) * 2


def cpu_step():
    opt.zero_grad()
    o = final(trunk(xc)) * keep
    o = torch.cat([o[:, :-8], torch.relu(o[:, -8:])], dim=1)
    crit(o, yc).backward()
    opt.step()


for _ in range(5):
    cpu_step()
t0, k = time.perf_counter(), 0
while time.perf_counter() - t0 < 10.0:
    cpu_step()
    k += 1
cpu = {"value": round(k * 1024 / (time.perf_counter() - t0), 1), "unit": "columns/s", "cores": torch.get_num_threads(),
       "kind": "port", "sample": f"{k} steps of batch 1024, fp32 torch-CPU"}
print(json.dumps({"metric": "training columns/sec", "workload": "online MLP 557->9x256->368 relu, prune 12, huber, Adam(torch)",
                  "dtype": "bf16", "data": "synthetic", "per_batch": res, "cpu_baseline": cpu}))
