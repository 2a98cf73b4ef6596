"""Development: fused chain time (profile_step) of 5-hidden-layer models of one width W at batch B, wide chain (W > 512 or forced
by a 640-wide first layer) against the tuned chain: does a stage's time follow its BYTES (stream-bound) or its number of column
PASSES (latency-bound)?  python tools/chainw_width_time.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for units in ((512,) * 5, (640,) * 5, (768,) * 5, (896,) * 5, (1024,) * 5, (768, 640, 512, 640, 640), (640, 512, 512, 512, 512)):
    m = MLPEmulator(units=units, activation="leakyrelu", optimizer="Adam", max_batch=B, seed=0)
    for _ in range(5):
        m.train_on_batch(x, y, 1e-3)
    agg = {}
    for r in range(20):
        for k, (ms, cnt) in m.profile_step(x, y, 1e-3).items():
            a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
    w = sum(a * b for a, b in zip((124,) + units, units + (128,))) + 128 * 128
    print(units, "B", B, "weights", w, {k: round(v[0] / 20 * 1e3, 1) for k, v in agg.items() if v[1]}, "chain us per Mweight",
          round(agg.get("chain_fb", [0])[0] / 20 * 1e3 / (w / 1e6), 1), flush=True)
    m.close()
