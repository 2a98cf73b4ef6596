"""Development (a -DCWD_WAVE_STAMPS build through CLIMSIM_HIP_LIB): when each of the 8 waves of a workgroup reaches the stage barriers of
the wide chain's forward half - clocks after the previous barrier, mean over workgroups."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["CS_CHAIN_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climsim_amd import _lib  # noqa: E402
from climsim_amd.mlp import MLPEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
UNITS = tuple(int(u) for u in sys.argv[2].split(",")) if len(sys.argv) > 2 else (768, 640, 512, 640, 640)
m = MLPEmulator(units=UNITS, activation="leakyrelu", optimizer="RAdam", max_batch=B, seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for _ in range(5):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
mp = (B + 127) // 128 * 128
half = (mp // 32) * 64
buf = np.zeros(2 * half, dtype=np.uint64)
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), 2 * half))
st = buf[:half].reshape(-1, 64).astype(np.int64)
st = st[st[:, 54] > 0]
widths = list(UNITS) + [128]
prev = st[:, 54]
for i, w in enumerate(widths[:6]):
    arr = st[:, 8 * i:8 * i + 8] - prev[:, None]
    rel = st[:, 48 + i] - prev
    print(f"stage {i} width {w:4d}: waves reach the barrier at", np.round(arr.mean(axis=0)).astype(int).tolist(), "| barrier passed", int(rel.mean()),
          "| last - first", int((arr.max(axis=1) - arr.min(axis=1)).mean()))
    prev = st[:, 48 + i]
m.close()
