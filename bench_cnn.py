#!/usr/bin/env python3
"""CNN training and prediction throughput (BASELINE.json config 3: depth 12, width 406, 60 levels, batch 512) on
one MI355X.  Secondary measurement - bench.py stays the MLP training metric.  Prints one JSON line.
usage: bench_cnn.py [batch] [steps]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from climsim_amd import build  # noqa: E402

build.build()
from climsim_amd.cnn import CNNEmulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = 16 * B
m = CNNEmulator(depth=12, channel_width=406, max_batch=B, trainable=True, init_seed=0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((N, 124), device="cuda", generator=g) - 0.5).contiguous()
y = (torch.rand((N, 128), device="cuda", generator=g) * 0.1).contiguous()
perm = torch.randperm(N, device="cuda", generator=g)
loss = torch.zeros(4, device="cuda")


def timed(fn, reps):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


dt_train = timed(lambda i: m.train_on_batch(x, y, 1e-4, row_idx=perm[(i % 16) * B:(i % 16 + 1) * B], loss=loss, x3d=0, y3d=0), K)
first = m._losses(loss.cpu().numpy(), B)["loss"]
dt_pred = timed(lambda i: m.predict(x[:B], as_numpy=False), K)
fwd = 1.584e9                  # forward FLOP per column
print(json.dumps({"metric": "CNN training columns/sec", "value": round(B / dt_train, 1), "unit": "columns/s", "batch": B,
                  "ms_per_step": round(dt_train * 1e3, 3), "train_tflops_algorithmic": round(3 * fwd * B / dt_train / 1e12, 1),
                  "predict_columns_per_s": round(B / dt_pred, 1), "predict_tflops_algorithmic": round(fwd * B / dt_pred / 1e12, 1),
                  "loss_after": first, "dtype": "bf16 operands, fp32 accumulate/master/Adam",
                  "note": "trunk convs 240x224 LDS-DMA tap-GEMMs (whole columns, one row tile per chunk shared by the taps, loader waves; K pad 416, N pad 448), conv wgrad 256x224 with taps folded into one axis; dropout 0.175, mae_adjusted, Adam"}))
