"""Print per-phase shader-clock deltas of the chain kernels (development aid; needs a GPU)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["CS_CHAIN_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climsim_amd import _lib  # noqa: E402
from climsim_amd.mlp import MLPEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
FLAGS = int(os.environ.get("CS_FLAGS", "0"))
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0, flags=FLAGS)
x = torch.randn(B, 124, device="cuda") * 0.2
y = torch.randn(B, 128, device="cuda") * 0.05
for _ in range(5):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
mp = (B + 127) // 128 * 128
half = (mp // 32) * 64                       # [2][m_pad_max / 32][64] stamps (fwd, bwd)
words = 2 * half
# clear stale stamps, run ONE step, read
buf = np.zeros(words, dtype=np.uint64)
agg = {}
for r in range(20):
    for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
us = {k: v[0] / 20 * 1e3 for k, v in agg.items() if v[1]}
print({k: round(v, 1) for k, v in us.items()})
m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), words))
for name, base, kind in (("fwd", 0, "chain_fwd"), ("bwd", half, "chain_bwd")):
    st = buf[base:base + half].reshape(-1, 64).astype(np.int64)
    st = st[st[:, 0] > 0]
    grid = st.shape[0]
    n = 1
    while n < 64 and np.all(st[:, n] >= st[:, n - 1]) and np.all(st[:, n] > 0):
        n += 1
    n = min(n, 60)
    d = np.diff(st[:, :n], axis=1)
    tot = st[:, n - 1] - st[:, 0]
    real = (st[:, 63] - st[:, 62]) / 100.0                      # us per workgroup (100 MHz)
    hw = st[:, 61] & 0xffffffff
    xcc = (st[:, 61] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    place = xcc * 1000 + se * 100 + sh * 10 + cu                 # sh may be unused on this part
    uniq, cnt = np.unique(place, return_counts=True)
    print(name, "grid", grid, "slots", n, "mean ticks per phase:", np.round(d.mean(axis=0)).astype(int).tolist())
    print("   per-WG ticks mean/min/max", int(tot.mean()), int(tot.min()), int(tot.max()),
          "| per-WG us mean/min/max", round(real.mean(), 1), round(real.min(), 1), round(real.max(), 1),
          "| kernel (events) us", round(us.get(kind, us.get("chain_fb", 0.0)), 1), "| ticks/us", round(float((tot / real).mean()), 1))
    print("   wall span us (100 MHz, all WGs)", (st[:, 63].max() - st[:, 62].min()) / 100.0, "start spread us", (st[:, 62].max() - st[:, 62].min()) / 100.0)
    print("   distinct (xcc,se,sh,cu) places", len(uniq), "max WGs on one place", int(cnt.max()), "WGs per xcc", np.bincount(xcc, minlength=8).tolist())
f = buf[:half].reshape(-1, 64).astype(np.int64); f = f[f[:, 0] > 0]
b = buf[half:].reshape(-1, 64).astype(np.int64); b = b[b[:, 0] > 0]
t0 = f[:, 62].min()
print("step timeline (us, 100 MHz clock): fwd first start 0, fwd last end", (f[:, 63].max() - t0) / 100.0,
      "| bwd first start", (b[:, 62].min() - t0) / 100.0, "bwd last end", (b[:, 63].max() - t0) / 100.0)
