"""Development: is the many-trials pool bound by the host (enqueue) or by the GPU?  Prints total time per trial step for K = 1, 8 and the host cost of one train_on_batch call."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from climsim_amd import build; build.build()
from climsim_amd.hpo import TrialPool
B=3072
g = torch.Generator(device="cuda").manual_seed(0)
n = 64 * B
x = (torch.rand((n, 124), device="cuda", generator=g) - 0.5).contiguous()
y = (torch.rand((n, 128), device="cuda", generator=g) * 0.1).contiguous()
for K in (1, 8):
    pool = TrialPool([dict(units=(512,) * 5, activation="leakyrelu", optimizer="Adam", batch_size=B)] * K)
    pool.fit(x, y, epochs=1, steps_per_epoch=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.fit(x, y, epochs=1, steps_per_epoch=64)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(K, "enqueue-return ms", round((t1 - t0) * 1e3, 2), "total ms", round((t2 - t0) * 1e3, 2), "per trial-step us", round((t2 - t0) / (64 * K) * 1e6, 1))
    pool.close()
# host cost of one step: the first calls after a synchronise return as soon as the three launches are enqueued
from climsim_amd.mlp import MLPEmulator
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
idx = torch.randperm(n, device="cuda")[:B]
for _ in range(5):
    m.train_on_batch(x, y, 1e-3, row_idx=idx)
torch.cuda.synchronize()
ts = []
for _ in range(12):
    t0 = time.perf_counter()
    m.train_on_batch(x, y, 1e-3, row_idx=idx)
    ts.append((time.perf_counter() - t0) * 1e6)
torch.cuda.synchronize()
print("host us per train_on_batch call (first 12 after a sync):", [round(t, 1) for t in ts])
