"""Development: per-phase shader-clock deltas of the cooperative chain (CS_CHAIN_DBG stamps)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["CS_CHAIN_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climsim_amd import _lib  # noqa: E402
from climsim_amd.mlp import MLPEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = MLPEmulator(units=(512,) * 5, max_batch=8192, seed=0, cooperative=True)
x = torch.randn(B, 124, device="cuda") * 0.2
y = torch.randn(B, 128, device="cuda") * 0.05
for _ in range(10):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
words = 2 * 256 * 64
buf = np.zeros(words, dtype=np.uint64)
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), words))
st = buf.reshape(256, 128).astype(np.int64)
st = st[st[:, 0] > 0]
print("workgroups", st.shape[0])
names = ["k-loop(w0)", "k-loop(all)", "epi+publish", "wait+gather"]
t0 = st[:, 0]
pos = 1
L = 7
rows = []
for i in range(2 * L - 1):
    last = i == 2 * L - 2
    k = 2 if last else 4
    seg = st[:, pos:pos + k]
    prev = st[:, pos - 1]
    d = np.diff(np.concatenate([prev[:, None], seg], axis=1), axis=1)
    rows.append((("fwd %d" % i) if i < L else ("bwd %d" % (i - L)), d.mean(axis=0).round().astype(int).tolist(), int(d.sum(axis=1).mean())))
    pos += k
for r in rows:
    print(r)
tot = st[:, pos - 1] - st[:, 0]
print("total ticks after prologue mean/min/max", int(tot.mean()), int(tot.min()), int(tot.max()), "| sum of stage means", sum(r[2] for r in rows))
