"""Development: shader-clock stamps of the wide chain (k_chainw_fb), wave 0 of every workgroup: k-loop / epilogue of each pass, stage
barriers.  Default = the continuous stream (one stamp pair per column tile of the wave); `CS_CHAINW_STREAM=0` = the per-pass form
(one pair per pass of two tiles)."""
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
STREAM = os.environ.get("CS_CHAINW_STREAM", "1") != "0"
FINE = os.environ.get("FINE", "0") == "1"          # a -DCWD_FINE_STAMPS build: prologue in three parts, heads in two
m = MLPEmulator(units=UNITS, activation="leakyrelu", optimizer="RAdam", max_batch=B, seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for _ in range(5):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
agg = {}
for r in range(20):
    for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
print(UNITS, "B", B, {k: round(v[0] / 20 * 1e3, 1) for k, v in agg.items() if v[1]})
mp = (B + 127) // 128 * 128
half = (mp // 32) * 64
buf = np.zeros(2 * half, dtype=np.uint64)
m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
_lib.check(m.lib.cs_mlp_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), 2 * half))


def passes(width):
    nt = width // 32
    t0 = nt // 8 + (1 if nt % 8 else 0)          # wave 0's tiles
    return t0 if STREAM else (t0 + 1) // 2


widths = list(UNITS) + [128]
fwd_w = widths                                    # hidden stages of the forward pass (the heads stage follows)
kp = [128] + [(w + 31) // 32 * 32 for w in widths]        # contraction widths (input 124 -> 128)
bwd_w = [128] + widths[::-1][1:]                  # Nc of the backward stages: heads' input, then down to layer 1's input
for name, base, ws in (("fwd", 0, fwd_w), ("bwd", half, bwd_w)):
    st = buf[base:base + half].reshape(-1, 64).astype(np.int64)
    st = st[st[:, 0] > 0]
    real = (st[:, 63] - st[:, 62]) / 100.0
    i = 1
    d = lambda a, b: float((st[:, b] - st[:, a]).mean())
    print(f"== {name}: {st.shape[0]} workgroups, {real.mean():.1f} us each (100 MHz clock); clocks, mean over workgroups")
    if FINE and name == "fwd":
        print(f"   prologue: queue primed {d(0, 1):.0f} | bias + row-index loads issued {d(1, 2):.0f} | landed {d(2, 3):.0f} | LDS, barrier {d(3, 4):.0f} | inputs landed {d(4, 5):.0f} | normalise, LDS, barrier {d(5, 6):.0f}")
        i = 6
    else:
        print(f"   prologue {d(0, 1):.0f}")
    tot = {"k-loop": 0.0, "epilogue": 0.0, "barrier": 0.0}
    for w in ws:
        np_ = passes(w)
        row = []
        for _ in range(np_):
            row.append((d(i, i + 1), d(i + 1, i + 2))); i += 2
            tot["k-loop"] += row[-1][0]; tot["epilogue"] += row[-1][1]
        bar = d(i, i + 1); i += 1; tot["barrier"] += bar
        print(f"   stage width {w:4d}: " + " ".join(f"[k-loop {a:.0f} epi {b:.0f}]" for a, b in row) + f" barrier {bar:.0f}")
    if name == "fwd" and FINE:
        print(f"   heads: k-loop {d(i, i + 1):.0f} | targets, loss, stores {d(i + 1, i + 2):.0f}"); i += 2
    elif name == "fwd":
        print(f"   heads {d(i, i + 1):.0f}"); i += 1
    print(f"   tail {d(i, i + 1):.0f}   total {d(0, i + 1):.0f}   sums " + " ".join(f"{k} {v:.0f}" for k, v in tot.items()))
m.close()
