"""Development: step time and per-kernel times of the cfg-MLP at a few batch sizes (plain chain, no cooperative launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
import ctypes
from climsim_amd import _lib
for B in [int(a) for a in sys.argv[1:]] or (1024, 3072, 4096, 8192, 16384):
    m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
    x = torch.randn(B, 124, device="cuda") * 0.2
    y = torch.randn(B, 128, device="cuda") * 0.05
    for _ in range(20):
        m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    with _lib.profile_session(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) as prof:
        for _ in range(40):
            m.train_on_batch(x, y, 1e-3)
    print(B, round(dt * 1e6, 1), {k: round(v[0] / 40 * 1e3, 1) for k, v in prof.times.items() if v[1]}, flush=True)
    m.close()
