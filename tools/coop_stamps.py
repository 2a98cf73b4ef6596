"""Per-phase s_memtime deltas of the cooperative chain (k_chain_coop_fb; development aid, needs a GPU).
Slots: 0 = prologue done; per stage: k-loop done (wave 0) | all waves (barrier) | epilogue + publish drained | gathered."""
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
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0, cooperative=True)
x = torch.randn(B, 124, device="cuda") * 0.2
y = torch.randn(B, 128, device="cuda") * 0.05
for _ in range(20):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
agg = {}
for r in range(20):
    for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
print({k: round(v[0] / 20 * 1e3, 1) for k, v in agg.items() if v[1]})
m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
words = 256 * 128
buf = np.zeros(words, dtype=np.uint64)
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), words))
st = buf.reshape(256, 128).astype(np.int64)
st = st[st[:, 0] > 0]
n = 1
while n < 128 and np.all(st[:, n] > 0) and np.all(st[:, n] >= st[:, n - 1]):
    n += 1
d = np.diff(st[:, :n], axis=1)
print("workgroups", st.shape[0], "stamps", n, "ticks from first to last stamp mean/max", int((st[:, n - 1] - st[:, 0]).mean()), int((st[:, n - 1] - st[:, 0]).max()))
names = ["k-loop (wave 0)", "all waves", "epilogue+drain", "gather"]
dm = d.mean(axis=0)
for s in range(0, n - 1, 4):
    print("stage %2d:" % (s // 4), "  ".join("%s %5d" % (names[j], int(dm[s + j])) for j in range(min(4, n - 1 - s))))
tot = {names[j]: int(sum(dm[s + j] for s in range(0, n - 1, 4) if s + j < n - 1)) for j in range(4)}
print("sums over the stages (ticks; ~2100 per us):", tot)
m.close()
