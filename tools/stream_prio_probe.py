#!/usr/bin/env python3
"""Probe (round 4): does a LOW-priority side stream let the loader kernel run in the gaps of the training step (k_wgrad3 leaves 37
of 256 CUs idle for 34 us of every 122 us step) instead of beside the layer chain?  Pass time of the streamed trainer, median of 5,
for: loader on the training stream | side stream, default priority | side stream at the lowest priority, training on a stream at
the highest.   python tools/stream_prio_probe.py [chunks] [timesteps] [batch]"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests", "golden"))
from climsim_amd import build  # noqa: E402
build.build()
from climsim_amd.assets import load_grid_info, load_npz_assets  # noqa: E402
from climsim_amd.data_utils import data_utils  # noqa: E402
from climsim_amd.loader import GpuColumnLoader  # noqa: E402
from climsim_amd.mlp import MLPEmulator  # noqa: E402
from climsim_amd.stream import StreamedTrainer  # noqa: E402

NCOL = 21600
G = os.path.join(R, "tests", "golden")
nch, T, B = ([int(v) for v in sys.argv[1:4]] + [8, 8, 8192])[:3] if len(sys.argv) > 3 else (8, 8, 8192)
grid = load_grid_info(os.path.join(G, "grid_lowres.npz"))
sets = [load_npz_assets(os.path.join(G, "norm_lowres.npz"), k) for k in ("input_mean", "input_max", "input_min", "output_scale")]
du = data_utils(grid, *sets, ml_backend="pytorch")
du.set_to_v1_vars()
ld = GpuColumnLoader(du)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
chunks = []
for c in range(nch):
    mli = ld._sub[None, :, None] + ld._div[None, :, None] * 0.15 * torch.randn((T, ld.n_in, NCOL), device=dev, dtype=torch.float64, generator=g)
    mlo = 0.05 * torch.randn((T, ld.n_out, NCOL), device=dev, dtype=torch.float64, generator=g) / ld._scale[None, :, None]
    mlo[:, :120] = mli[:, :120] + 1200.0 * mlo[:, :120]
    mlo[:, 120:] = mlo[:, 120:].abs()
    chunks.append((mli.contiguous(), mlo.contiguous()))
least, greatest = torch.cuda.Stream.priority_range()
print("stream priority range (least, greatest):", least, greatest)
model = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
hi = torch.cuda.Stream(device=dev, priority=greatest)
variants = {"main": (dict(loader_on="main"), None), "side": (dict(loader_on="side"), None),
            "side low / train default": (dict(loader_on="side", side_priority=least), None),
            "side low / train high": (dict(loader_on="side", side_priority=least), hi),
            "side default / train high": (dict(loader_on="side"), hi)}
trainers = {k: (StreamedTrainer(model, ld, batch_size=B, slots=2, **kw), s) for k, (kw, s) in variants.items()}


def one(st, s):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if s is None:
        st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=1)
    else:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for st, s in trainers.values():
    one(st, s)
times = {k: [] for k in trainers}
for _ in range(5):
    for k, (st, s) in trainers.items():
        times[k].append(one(st, s))
rows = nch * T * NCOL
for k, v in times.items():
    m = sorted(v)[len(v) // 2]
    print(f"{k:28s} median {m:7.2f} ms = {rows / m / 1e3:6.2f} M columns/s   passes {[round(x, 2) for x in v]}")
model.close()
