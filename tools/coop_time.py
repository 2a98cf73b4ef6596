"""Development: step time of the cfg-MLP with and without the cooperative chain at small batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
import ctypes
from climsim_amd import _lib
for B in (256, 512, 1024, 1536, 2048):
    out = {}
    for coop in (False, True):
        m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0, cooperative=coop)
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
        m.get_weights()
        out[coop] = (round(dt * 1e6, 1), {k: round(v[0] / 40 * 1e3, 1) for k, v in prof.times.items() if v[1]})
        m.close()
    print(B, "plain", out[False], "| coop", out[True], "| speedup", round(out[False][0] / out[True][0], 2), flush=True)
