#!/usr/bin/env python3
"""Many trials per GPU: aggregate training columns/s of K independent cfg-MLP trials on one MI355X
against one trial alone, at a batch size of the reference's search space (hpo_baseline_v1.py:66-74: 48..3072; default 1024).
Two forms: `grouped` - ONE launch per kernel kind for all K trials (cs_mlp_group_*) - and `streams` - K engines on K HIP
streams stepped round-robin (round 1's form).  Also: eight architectures drawn from the reference's search space, and the
RPN ensemble shape (32 members of 124-768-640-512-640-640-128, rpn_model_v1_data.py:71-163).  One JSON line.

    python bench_hpo.py [batch] [steps]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.hpo import TrialPool, sample_trial  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator(device="cuda").manual_seed(0)
n = max((STEPS + 10) * B, 65536)                                    # an epoch of STEPS distinct batches per trial
x = (torch.rand((n, 124), device="cuda", generator=g) - 0.5).contiguous()
y = (torch.rand((n, 128), device="cuda", generator=g) * 0.1).contiguous()


def rate(trials, grouped, steps=STEPS):
    pool = TrialPool(trials, grouped=grouped)
    pool.fit(x, y, epochs=1, steps_per_epoch=10)                      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.fit(x, y, epochs=1, steps_per_epoch=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cols = sum(int(t["batch_size"]) for t in trials) * steps
    pool.close()
    return cols / dt, dt / steps


cfg = dict(units=(512,) * 5, activation="leakyrelu", optimizer="Adam", batch_size=B)
res = {"grouped": {}, "streams": {}}
step_ms = {}
for K in (1, 2, 4, 8, 16, 32):
    r, dt = rate([cfg] * K, True)
    res["grouped"][K] = round(r, 1)
    step_ms[K] = round(dt * 1e3, 4)
    if K <= 16:
        res["streams"][K] = round(rate([cfg] * K, False)[0], 1)
one = res["streams"][1]

# where a grouped step of 8 trials spends its time: event pairs around its three launches (cs_mlp_group_profile_step)
pool = TrialPool([cfg] * 8, grouped=True)
pool.fit(x, y, epochs=1, steps_per_epoch=5)
grp = pool.groups[0]
import ctypes  # noqa: E402
from climsim_amd import _lib  # noqa: E402
idxs = [[torch.randint(0, n, (B,), device="cuda", generator=g) for _ in range(8)] for _ in range(20)]
with _lib.profile_session(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) as prof:     # 20 back-to-back grouped steps
    for idx in idxs:
        grp.train_on_batch(x, y, 1e-3, row_idx=idx)
agg = {kind: ms / 20 for kind, (ms, cnt) in prof.times.items() if cnt}
kernels_k8 = {k: round(v * 1e3, 1) for k, v in agg.items()}
pool.close()

# eight architectures drawn from the reference's search space (widths 128..1024, 2..12 layers; fixed seed)
rng = np.random.default_rng(7)
mix = []
while len(mix) < 8:
    t = sample_trial(rng)
    t["batch_size"] = B
    mix.append(t)
seq = sum(1.0 / rate([t], False, steps=50)[0] * B * 50 for t in mix)          # seconds, one after the other
conc_g, _ = rate(mix, True, steps=50)
conc_s, _ = rate(mix, False, steps=50)
res_mix = {"trials": [{"units": list(t["units"]), "activation": t["activation"], "optimizer": t["optimizer"]} for t in mix],
           "sequential_columns_per_s": round(8 * 50 * B / seq, 1), "grouped_columns_per_s": round(conc_g, 1),
           "streams_columns_per_s": round(conc_s, 1), "speedup_grouped": round(conc_g * seq / (8 * 50 * B), 2),
           "speedup_streams": round(conc_s * seq / (8 * 50 * B), 2)}

# RPN's ensemble: 32 members of one wide shape, each on its own batch
rpn = dict(units=(768, 640, 512, 640, 640), activation="leakyrelu", optimizer="Adam", batch_size=B)
rpn_one, _ = rate([rpn], False, steps=50)
rpn_32, rpn_dt = rate([rpn] * 32, True, steps=50)
res_rpn = {"members": 32, "units": list(rpn["units"]), "one_member_columns_per_s": round(rpn_one, 1),
           "ensemble_columns_per_s": round(rpn_32, 1), "speedup": round(rpn_32 / rpn_one, 2), "ms_per_ensemble_step": round(rpn_dt * 1e3, 4)}

print(json.dumps({"metric": "aggregate training columns/sec of K concurrent trials", "unit": "columns/s", "batch": B,
                  "model": "cfg-MLP 5x512", "columns_per_s_by_K": res, "ms_per_grouped_step_by_K": step_ms,
                  "grouped_step_us_by_kernel_K8": kernels_k8, "speedup_vs_one": {f: {k: round(v / one, 2) for k, v in d.items()} for f, d in res.items()},
                  "search_space_mix": res_mix, "rpn_ensemble": res_rpn}))
