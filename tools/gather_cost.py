"""Development (round 6): what the in-kernel gather of a step's rows costs the layer chain at 8192 columns - the same step on
(a) the same 8192 contiguous rows every step (hot in the L2s / memory-side cache, no row indices), (b) contiguous rows walking through a
1 M-row split, (c) rows gathered through a device permutation of the 1 M-row split (what fit() and bench.py do)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
from climsim_amd import _lib
B, N = 8192, 1 << 20
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn((N, 124), device="cuda", generator=g) * 0.2).contiguous()
y = (torch.randn((N, 128), device="cuda", generator=g) * 0.05).contiguous()
perm = torch.randperm(N, device="cuda", generator=g)
def run(kind, steps=300):
    def step(i):
        if kind == "hot":
            m.train_on_batch(x[:B], y[:B], 1e-3)
        elif kind == "walk":
            lo = (i % (N // B)) * B
            m.train_on_batch(x[lo:lo + B], y[lo:lo + B], 1e-3)
        else:
            lo = (i % (N // B)) * B
            m.train_on_batch(x, y, 1e-3, row_idx=perm[lo:lo + B])
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    with _lib.profile_session(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) as prof:
        for i in range(40):
            step(i)
    return round(dt * 1e6, 1), {k: round(v[0] / 40 * 1e3, 1) for k, v in prof.times.items() if v[1]}
for r in range(3):
    for kind in ("hot", "walk", "gather"):
        print("rot", r, kind, *run(kind), flush=True)
