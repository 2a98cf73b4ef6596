"""Where a streamed pass spends its time (development aid; needs a GPU): HIP events on the training stream around every phase of
every chunk of StreamedTrainer.fit_chunks - loader launch, permutation, loss-slot allocation, the steps - against the same phases of
"loader first, then training" on the same rows.  python tools/stream_stamps.py [chunks] [timesteps] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_stream  # noqa: E402  (builds the library)
from climsim_amd.assets import load_grid_info, load_npz_assets  # noqa: E402
from climsim_amd.data_utils import data_utils  # noqa: E402
from climsim_amd.loader import GpuColumnLoader  # noqa: E402
from climsim_amd.mlp import MLPEmulator  # noqa: E402

nch, T, B = (int(v) for v in (sys.argv[1:4] + ["4", "8", "8192"][len(sys.argv) - 1:]))
G = bench_stream.G
grid = load_grid_info(os.path.join(G, "grid_lowres.npz"))
sets = [load_npz_assets(os.path.join(G, "norm_lowres.npz"), k) for k in ("input_mean", "input_max", "input_min", "output_scale")]
du = data_utils(grid, *sets, ml_backend="pytorch")
du.set_to_v1_vars()
ld = GpuColumnLoader(du)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
chunks = []
for c in range(nch):
    mli = ld._sub[None, :, None] + ld._div[None, :, None] * 0.15 * torch.randn((T, ld.n_in, 21600), device=dev, dtype=torch.float64, generator=g)
    mlo = 0.05 * torch.randn((T, ld.n_out, 21600), device=dev, dtype=torch.float64, generator=g) / ld._scale[None, :, None]
    mlo[:, :120] = mli[:, :120] + 1200.0 * mlo[:, :120]
    chunks.append((mli.contiguous(), mlo.contiguous()))
model = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0)
gen = torch.Generator(device=dev).manual_seed(1)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def one_pass(tag):
    rows = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for mli, mlo in chunks:
        e0 = ev()
        x, y = ld.stack_raw(mli, mlo)
        e1 = ev()
        perm = torch.randperm(x.shape[0], device=dev, generator=gen)
        e2 = ev()
        sums = torch.zeros(((x.shape[0] + B - 1) // B, 2), dtype=torch.float32, device=dev)
        e3 = ev()
        h0 = time.perf_counter()
        for k, lo in enumerate(range(0, x.shape[0], B)):
            model.train_on_batch(x, y, 1e-3, row_idx=perm[lo:lo + B], loss=sums[k])
        h1 = time.perf_counter()
        e4 = ev()
        rows.append((e0, e1, e2, e3, e4, h1 - h0, k + 1))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(tag, "wall ms", round(wall * 1e3, 2), "columns/s", round(nch * T * 21600 / wall / 1e6, 2), "M")
    for i, (e0, e1, e2, e3, e4, host, steps) in enumerate(rows):
        print("  chunk", i, "loader %.3f perm %.3f zeros %.3f steps %.3f ms (%d steps, %.1f us each; host enqueue %.3f ms)" % (
            e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4), steps, e3.elapsed_time(e4) / steps * 1e3, host * 1e3))


one_pass("warm-up")
one_pass("loader + steps in stream order")
# the trainer itself, and the two halves back to back
from climsim_amd.stream import StreamedTrainer  # noqa: E402
st = StreamedTrainer(model, ld, batch_size=B, slots=2)
st.fit_chunks(iter(chunks[:2]), learning_rate=1e-3)
for carry in (True, False):
    st = StreamedTrainer(model, ld, batch_size=B, slots=2, carry_remainder=carry)
    st.fit_chunks(iter(chunks[:2]), learning_rate=1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=1)
    torch.cuda.synchronize(); print("StreamedTrainer.fit_chunks carry=%s wall ms" % carry, round((time.perf_counter() - t0) * 1e3, 2), "steps", r["steps"])
    st.trace = []
    st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=1)
    torch.cuda.synchronize()
    tr = st.trace
    print("   phases (ms since the previous mark):", " | ".join("%s %.3f" % (tr[i][0], tr[i - 1][1].elapsed_time(tr[i][1])) for i in range(1, len(tr))))
for nb in (768, 3072, 8192, 3072, 768):
    xx, yy = xs_probe = ld.stack_raw(*chunks[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        model.train_on_batch(xx[:nb], yy[:nb], 1e-3)
    torch.cuda.synchronize(); print("10 steps of %d rows: %.1f us each" % (nb, (time.perf_counter() - t0) * 1e5))
xs = [ld.stack_raw(a, b) for a, b in chunks]
torch.cuda.synchronize(); t0 = time.perf_counter()
for x, y in xs:
    perm = torch.randperm(x.shape[0], device=dev, generator=gen)
    for lo in range(0, x.shape[0], B):
        model.train_on_batch(x, y, 1e-3, row_idx=perm[lo:lo + B])
torch.cuda.synchronize(); print("train only wall ms", round((time.perf_counter() - t0) * 1e3, 2))
torch.cuda.synchronize(); t0 = time.perf_counter()
for x, y in xs:
    for lo in range(0, x.shape[0] - B + 1, B):
        model.train_on_batch(x, y, 1e-3, row_idx=None, n=B)
torch.cuda.synchronize(); print("train only, full batches, no permutation: wall ms", round((time.perf_counter() - t0) * 1e3, 2))
