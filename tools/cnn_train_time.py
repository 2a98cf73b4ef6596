"""Development: training step time of the published CNN shape (env knobs apply)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.cnn import CNNEmulator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = CNNEmulator(depth=12, channel_width=406, max_batch=B, trainable=True, init_seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for _ in range(3):
    m.train_on_batch(x, y, 1e-4, x3d=0, y3d=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    m.train_on_batch(x, y, 1e-4, x3d=0, y3d=0)
torch.cuda.synchronize()
print("B", B, "splits", os.environ.get("CS_CNN_WGRAD_SPLITS", "auto"), "ms/step", round((time.perf_counter() - t0) / 20 * 1e3, 3))
