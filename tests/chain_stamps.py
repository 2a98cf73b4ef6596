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
words = 2 * (mp // 64) * 64
buf = np.zeros(words, dtype=np.uint64)
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), words))
bm = 128 if FLAGS & 8 else 64
grid = mp // bm
for name, base in (("fwd", 0), ("bwd", (mp // 64) * 64)):
    st = buf[base:base + grid * 64].reshape(grid, 64).astype(np.int64)
    n = int((st[0] > 0).sum())
    d = np.diff(st[:, :n], axis=1)
    print(name, "grid", grid, "slots", n, "mean ticks per phase:", np.round(d.mean(axis=0)).astype(int).tolist(),
          "total", int((st[:, n - 1] - st[:, 0]).mean()), "span(all WGs)", int(st[:, :n].max() - st[:, :n].min()))
