"""Development: training step time of the published MLP (lot-147/trial_0027: 768,640,512,640,640, RAdam, batch 3072)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
for units, B in (((768, 640, 512, 640, 640), 3072), ((768, 640, 512, 640, 640), 8192), ((512,) * 5, 3072)):
    m = MLPEmulator(units=units, activation="leakyrelu", optimizer="RAdam", max_batch=B, seed=0)
    x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
    y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
    for _ in range(5):
        m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(units, "B", B, "ms/step", round(dt * 1e3, 4), "Mcol/s", round(B / dt / 1e6, 2), "params", m.count_params())
    m.close()
m = MLPEmulator(units=(768, 640, 512, 640, 640), activation="leakyrelu", optimizer="RAdam", max_batch=3072, seed=0)
x = (torch.rand((3072, 124), device="cuda") - 0.5).contiguous(); y = (torch.rand((3072, 128), device="cuda") * 0.1).contiguous()
agg = {}
for r in range(20):
    for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
print({k: (round(v[0] / 20 * 1e3, 1), v[1] / 20) for k, v in agg.items() if v[1]})
