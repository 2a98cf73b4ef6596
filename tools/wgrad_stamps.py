"""Phases of k_wgrad3 per workgroup (development aid; needs a GPU, CS_CHAIN_DBG stamps - csrc/wgrad2.h).  usage: wgrad_stamps.py [batch]"""
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
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
x = torch.randn(B, 124, device="cuda") * 0.2
y = torch.randn(B, 128, device="cuda") * 0.05
for _ in range(10):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
agg = {}
for r in range(20):
    for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
print({k: round(v[0] / 20 * 1e3, 1) for k, v in agg.items() if v[1]})
m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
grid = C.c_int32(0)
_lib.check(m.lib.cs_mlp_debug_stamps_wgrad(m._h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(grid)))
st = buf[:grid.value * 8].reshape(-1, 8).astype(np.int64)
t0 = st[:, 5].min()
start = (st[:, 5] - t0) / 100.0
end = (st[:, 6] - t0) / 100.0
span = end - start
tick = float(np.median((st[:, 4] - st[:, 0]) / np.maximum(span, 1e-9)))
d = np.diff(st[:, :5], axis=1) / tick
names = ["entry -> stage 0 landed (set-up, ring fill)", "contraction (loop)", "result: LDS staging + store issue", "stores acknowledged"]
print(f"k_wgrad3 at {B} columns: {grid.value} workgroups, {int(st[:, 7].min())}-{int(st[:, 7].max())} stages each, ~{tick:.0f} shader clocks per us")
print(f"kernel span (first entry -> last exit) {end.max():.1f} us; entries spread over {start.max():.2f} us; exits {end.min():.1f}-{end.max():.1f} us; per workgroup {span.mean():.1f} us mean")
for i, nme in enumerate(names):
    print(f"  {nme:46s} {d[:, i].mean():6.2f} us mean ({d[:, i].min():.2f}-{d[:, i].max():.2f})")
print(f"  loop per stage: {(d[:, 1] * tick / np.maximum(st[:, 7], 1)).mean():.0f} clocks (a 64-row stage = 16 MFMAs of 32 clocks per compute wave = 512)")
