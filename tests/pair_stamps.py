"""Development: per-phase shader-clock deltas of the pair chain (csrc/chain2.h, CS_CHAIN_DBG stamps)."""
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
cw, hw = st[:, :64], st[:, 64:]
L = 7
# publishing stages: forward 0..4, backward 1..4 (of 6)
stages = [("fwd %d" % i, i <= 4) for i in range(L)] + [("bwd %d" % i, 1 <= i <= 4) for i in range(L - 1)]
print("prologue", int((cw[:, 1] - cw[:, 0]).mean()))
cp, hp = 2, 2
for si, (name, pub) in enumerate(stages):
    if name == "bwd 0":
        print("loss flush + transition", int((cw[:, cp] - cw[:, cp - 1]).mean()))
        cp += 1; hp += 1
    kloop = cw[:, cp] - cw[:, cp - 1]
    epi = cw[:, cp + 1] - cw[:, cp]
    line = "%s k-loop %6d  epilogue %5d" % (name, kloop.mean(), epi.mean())
    b2 = hw[:, hp + 1]
    if pub:
        line += "  | exchange after B2: stores acknowledged %5d, partner flag %5d, gathered %5d" % (
            (hw[:, hp + 2] - b2).mean(), (hw[:, hp + 3] - b2).mean(), (hw[:, hp + 4] - b2).mean())
        hp += 3
    print(line)
    cp += 2; hp += 2
print("total", int((cw[:, cp] - cw[:, 0]).mean()), "max", int((cw[:, cp] - cw[:, 0]).max()))
