"""Development: model.predict throughput of the cfg-MLP against the chunk size (python tools/predict_time.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
N = 1 << 20
x = torch.randn(N, 124, device="cuda") * 0.2
for B in (8192, 16384, 32768, 65536, 131072):
    m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
    m.predict(x, as_numpy=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        m.predict(x, as_numpy=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print(B, "median ms", round(t * 1e3, 3), "M columns/s", round(N / t / 1e6, 1), flush=True)
    m.close()
