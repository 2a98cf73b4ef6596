"""Phases of the conv weight-gradient kernel k_conv_wgrad2l per workgroup and queue entry (development aid; needs a GPU, CS_CNN_DBG
stamps - csrc/conv_wgrad2.h).  usage: cnn_wgrad_stamps.py [batch]      - depth 12, width 406 (BASELINE config 3)"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["CS_CNN_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climsim_amd import _lib, build  # noqa: E402

build.build()
from climsim_amd.cnn import CNNEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SLOTS = 64
m = CNNEmulator(depth=12, channel_width=406, max_batch=B, trainable=True, init_seed=0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((B, 124), device="cuda", generator=g) - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda", generator=g) * 0.1).contiguous()
loss = torch.zeros(4, device="cuda")
for _ in range(5):
    m.train_on_batch(x, y, 1e-4, loss=loss, x3d=0, y3d=0)
torch.cuda.synchronize()
buf = np.zeros(2048 * SLOTS, dtype=np.uint64)
grid = C.c_int32(0)
_lib.check(m.lib.cs_cnn_debug_stamps(m._h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(grid)))
st = buf[:grid.value * SLOTS].reshape(-1, SLOTS)
TOP = np.uint64(1) << np.uint64(63)
MASK = (np.uint64(1) << np.uint64(48)) - np.uint64(1)
rows = []            # (wg, entry no, slabs, start, landed, loop done, flush issued) in shader clocks
wg = []              # (entry us, exit us, entries, xcc)
for b in range(st.shape[0]):
    w = st[b]
    if w[0] == 0:
        continue
    end = [i for i in range(2, SLOTS) if w[i] & TOP]
    if not end:
        continue                                  # more entries than stamp slots: not summarised
    e = end[0]
    n = (e - 2) // 4
    for i in range(n):
        a = w[2 + 4 * i: 6 + 4 * i]
        rows.append((b, i, int(a[0] >> np.uint64(48)), int(a[0] & MASK), int(a[1] & MASK), int(a[2] & MASK), int(a[3] & MASK), int(a[1] >> np.uint64(48))))
    wg.append((int(w[0]), int(w[e] & ~TOP), n, int((w[1] >> np.uint64(32)) & np.uint64(0xf))))
rows = np.array(rows, dtype=np.int64)
wg = np.array(wg, dtype=np.int64)
t0 = wg[:, 0].min()
span = (wg[:, 1] - wg[:, 0]) / 100.0
kernel = (wg[:, 1].max() - t0) / 100.0
clk_total = np.zeros(len(wg))
# shader clocks per us from workgroups with >= 1 entry: (last flush stamp - first start) against the 100 MHz span
first = {}
last = {}
for r in rows:
    first.setdefault(r[0], r[3]); last[r[0]] = r[6]
ticks = np.array([last[b] - first[b] for b in first])
tick = float(np.median(ticks / np.maximum(span[:len(ticks)], 1e-9))) if len(ticks) else 2100.0
fill = (rows[:, 4] - rows[:, 3]) / tick
loop = (rows[:, 5] - rows[:, 4]) / tick
flush = (rows[:, 6] - rows[:, 5]) / tick
slabs = rows[:, 2]
print(f"k_conv_wgrad2l at batch {B}: {len(wg)} workgroups, {len(rows)} queue entries ({wg[:, 2].min()}-{wg[:, 2].max()} per workgroup), "
      f"{slabs.min()}-{slabs.max()} slabs of 32 rows per entry, ~{tick:.0f} shader clocks per us")
print(f"kernel span (first entry -> last exit) {kernel:.1f} us; workgroups: last START at {(wg[:, 0].max() - t0) / 100.0:.1f} us, exits {((wg[:, 1] - t0) / 100.0).min():.1f}-{((wg[:, 1] - t0) / 100.0).max():.1f} us, "
      f"mean busy {span.mean():.1f} us = {span.mean() / kernel:.3f} of the span")
tot = fill.sum() + loop.sum() + flush.sum()
for nme, v in (("entry start -> slab 0 landed (set-up, ring fill)", fill), ("loop", loop), ("flush issued (not acknowledged)", flush)):
    print(f"  {nme:50s} {v.mean():7.2f} us mean per entry ({v.min():.2f}-{v.max():.2f}) = {100 * v.sum() / tot:4.1f} %")
per_slab = (rows[:, 5] - rows[:, 4]) / np.maximum(slabs, 1)
print(f"  loop: {per_slab.mean():.0f} clocks per slab mean ({per_slab.min():.0f}-{per_slab.max():.0f}); the slab's 28 MFMAs per wave x 2 waves per SIMD = 896 clocks")
for kind, label in ((6, "3-tap tiles (tap-shared X tile, 5 slots)"), (16, "1-tap tiles (16 windows, 4 slots)")):     # round 6: k_conv_wgrad3l tags the tile type
    sel = rows[:, 7] == kind
    if sel.any():
        print(f"  {label}: {int(sel.sum())} entries, {per_slab[sel].mean():.0f} clocks per slab mean ({per_slab[sel].min():.0f}-{per_slab[sel].max():.0f}), "
              f"set-up + fill {fill[sel].mean():.2f} us, flush {flush[sel].mean():.2f} us")
print(f"  sum of loops / (workgroups x span) = {loop.sum() / (len(wg) * kernel):.3f}; slabs in all {int(slabs.sum())}")
for x in range(8):
    sel = wg[:, 3] == x
    if sel.any():
        print(f"  XCC {x}: {int(sel.sum()):3d} workgroups, {int(wg[sel, 2].sum()):4d} entries, exits {((wg[sel, 1] - t0) / 100.0).min():7.1f}-{((wg[sel, 1] - t0) / 100.0).max():7.1f} us")
