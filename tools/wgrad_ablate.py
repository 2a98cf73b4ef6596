"""Development: k_wgrad3 time with the result flush / the contraction loop removed (CS_WGRAD_ABLATE)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
x = torch.randn(B, 124, device="cuda") * 0.2
y = torch.randn(B, 128, device="cuda") * 0.05
for _ in range(5):
    m.profile_step(x, y, 0.0)
agg = {}
for r in range(30):
    for k, (ms, cnt) in m.profile_step(x, y, 0.0).items():
        a = agg.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
print(B, os.environ.get("CS_WGRAD_ABLATE", "0"), os.environ.get("CS_WGRAD_SPLITK", "auto"), {k: round(v[0] / 30 * 1e3, 1) for k, v in agg.items() if v[1]})
