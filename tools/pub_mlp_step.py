"""Development: the published MLP (768-640-512-640-640, LeakyReLU, RAdam) at its batch 3072, N steps - a single-shape run for
`rocprofv3 --kernel-trace --stats -- python3 tools/pub_mlp_step.py` (profiles/r06_rocprofv3_kernel_stats_pub_mlp.csv)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = MLPEmulator(units=(768, 640, 512, 640, 640), activation="leakyrelu", optimizer="RAdam", max_batch=B, seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for _ in range(20):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    m.train_on_batch(x, y, 1e-3)
torch.cuda.synchronize()
print("published MLP, batch", B, "ms/step", round((time.perf_counter() - t0) / N * 1e3, 4))
m.close()
