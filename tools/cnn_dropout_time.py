import sys, time, os, torch
sys.path.insert(0, os.getcwd())
from climsim_amd.cnn import CNNEmulator
B=512
for dr in (0.175, 0.0):
    m = CNNEmulator(depth=12, channel_width=406, max_batch=B, trainable=True, init_seed=0, dropout=dr)
    x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous(); y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
    for _ in range(3): m.train_on_batch(x, y, 1e-4, x3d=0, y3d=0)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): m.train_on_batch(x, y, 1e-4, x3d=0, y3d=0)
    torch.cuda.synchronize(); print("dropout", dr, "ms/step", round((time.perf_counter()-t0)/20*1e3,3)); m.close()
