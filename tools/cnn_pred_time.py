"""Development: prediction time per batch of the published CNN shape (env CS_CONV_ABLATE applies)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.cnn import CNNEmulator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = CNNEmulator(depth=12, channel_width=406, max_batch=B, init_seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
for _ in range(3):
    m.predict(x, as_numpy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    m.predict(x, as_numpy=False)
torch.cuda.synchronize()
print("B", B, "ablate", os.environ.get("CS_CONV_ABLATE", "0"), "ms/batch", round((time.perf_counter() - t0) / 20 * 1e3, 3))
