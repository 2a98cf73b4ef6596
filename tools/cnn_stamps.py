"""Development: phase stamps of k_conv3 (CS_CNN_DBG=<file>, 100 MHz) for one training step at batch 512 - per conv of the last
forward / backward / prediction program: wait at the pass barrier X, main loop, epilogue (compute wave 0) and the loader's pass
boundary (flag polls + row-tile requests, landing, barrier)."""
import os
import sys

import numpy as np

path = "/tmp/cnn_stamps.bin"
os.environ["CS_CNN_DBG"] = path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from climsim_amd.cnn import CNNEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = CNNEmulator(depth=12, channel_width=406, max_batch=B, trainable=True, init_seed=0)
x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous()
y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
for _ in range(3):
    m.train_on_batch(x, y, 1e-4, x3d=0, y3d=0)
m.predict(x, as_numpy=False)
torch.cuda.synchronize()
m.close()
st = np.fromfile(path, dtype=np.uint64).astype(np.int64).reshape(3, 1024, 2, 128)
for mode, name in ((1, "train fwd"), (2, "bwd"), (0, "predict")):
    c = st[mode, :, 0, :]; l = st[mode, :, 1, :]
    live = c[:, 0] > 0
    c = c[live]; l = l[live]
    n = int((c[0] > 0).sum())
    if n < 4:
        continue
    npass = (n - 1) // 3
    t0 = c[:, 0:3 * npass:3]; t1 = c[:, 1:3 * npass:3]; t2 = c[:, 2:3 * npass:3]
    nxt = np.concatenate([t0[:, 1:], c[:, 3 * npass:3 * npass + 1]], axis=1)
    if os.environ.get("BRIEF"):
        print(f"{name:10s} total {round(float((c[:, 3 * npass] - c[:, 0]).mean()) / 100, 1)} X {round(float((t1 - t0).sum(1).mean()) / 100, 1)} loop {round(float((t2 - t1).sum(1).mean()) / 100, 1)} epilogue {round(float((nxt - t2).sum(1).mean()) / 100, 1)} | loop per pass {np.round((t2 - t1).mean(0) / 100, 1).tolist()[3:6]} epi {np.round((nxt - t2).mean(0) / 100, 1).tolist()[3:6]}")
        continue
    print(f"== {name}: {c.shape[0]} workgroups, {npass} passes; us per pass, mean over workgroups (10 ns ticks)")
    print("   X wait  ", np.round((t1 - t0).mean(0) / 100, 1).tolist())
    print("   loop    ", np.round((t2 - t1).mean(0) / 100, 1).tolist())
    print("   epilogue", np.round((nxt - t2).mean(0) / 100, 1).tolist())
    print("   total us", round(float((c[:, 3 * npass] - c[:, 0]).mean()) / 100, 1), " sums: X", round(float((t1 - t0).sum(1).mean()) / 100, 1),
          "loop", round(float((t2 - t1).sum(1).mean()) / 100, 1), "epilogue", round(float((nxt - t2).sum(1).mean()) / 100, 1))
    nl = int((l[0] > 0).sum()) // 4
    a = l[:, 0:4 * nl:4]; b = l[:, 1:4 * nl:4]; cc = l[:, 2:4 * nl:4]; d = l[:, 3:4 * nl:4]
    print("   loader: polls + tile requests", np.round((b - a).mean(0) / 100, 1).tolist())
    print("   loader: landing              ", np.round((cc - b).mean(0) / 100, 1).tolist())
    print("   loader: wait at X            ", np.round((d - cc).mean(0) / 100, 1).tolist())
