"""N>1 path on CPU: two gloo processes run climsim_amd.dp.DataParallel with an oracle-backed
stand-in engine and must reproduce single-process training on the global batch (gradient sums are
additive over the round-robin shards; scale 1/(128*global_batch) is applied once)."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from climsim_amd.dp import DataParallel, shard_of_batch  # noqa: E402
from oracle import mlp_oracle as O  # noqa: E402

CFG = O.MLPConfig(hidden=(128, 128))
N_ROWS, GLOBAL_BATCH, STEPS = 1024, 256, 3


class OracleEngine:
    """CPU stand-in with the engine protocol (loss_grads / gradient_tensor / apply_gradients)."""

    def __init__(self, ws):
        self.ws = [w.copy() for w in ws]
        self.opt = O.Optimizer("Adam")
        self.grad = torch.zeros(sum(w.size for w in ws), dtype=torch.float32)
        self.shapes = [w.shape for w in ws]

    def gradient_tensor(self):
        return self.grad

    def get_weights(self):
        return [w.copy() for w in self.ws]

    def set_weights(self, ws):
        self.ws = [np.asarray(w, np.float32).copy() for w in ws]

    def loss_grads(self, x, y, row_idx=None, loss=None, normalise=False):
        idx = row_idx.numpy()
        l, _, g, _ = O.loss_and_grads(self.ws, x[idx], y[idx], CFG)
        n = len(idx)
        # engine contract: UNSCALED sums (d sum-of-squares / d param)
        self.grad.copy_(torch.from_numpy(np.concatenate([a.ravel() for a in g]) * np.float32(128 * n)))
        return l

    def apply_gradients(self, lr, grad_scale):
        flat = self.grad.numpy() * np.float32(grad_scale)
        gs, at = [], 0
        for s in self.shapes:
            k = int(np.prod(s))
            gs.append(flat[at:at + k].reshape(s))
            at += k
        self.ws = self.opt.apply(self.ws, gs, lr)


def _worker(rank, world, port, out_dir, payload="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = O.synth_columns(N_ROWS, seed=3)
    ws = O.glorot_init(CFG, seed=rank)          # ranks start DIFFERENT: broadcast must fix it
    eng = OracleEngine(ws)
    dp = DataParallel(eng, dist, grad_payload=payload)
    if payload == "bf16":                       # the collective alone: bf16(bf16(a) + bf16(b)), the same on both ranks
        v = torch.Generator().manual_seed(100 + rank)
        eng.grad.copy_(torch.randn(eng.grad.numel(), generator=v) * 3.0)
        dp.all_reduce_grads()
        np.save(os.path.join(out_dir, f"sum{rank}.npy"), eng.grad.numpy())
    dp.broadcast_weights()
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(N_ROWS, generator=g)
    for s in range(STEPS):
        dp.train_step(x, y, perm, s, GLOBAL_BATCH, 1e-3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), *eng.ws)
    dist.destroy_process_group()


def test_shards_partition_the_global_batch():
    perm = torch.randperm(1000, generator=torch.Generator().manual_seed(0))
    for world in (1, 2, 4, 8):
        parts = [shard_of_batch(perm, 2, 64, r, world) for r in range(world)]
        got = torch.cat(parts).sort().values
        want = perm[2 * 64:3 * 64].sort().values
        assert torch.equal(got, want)
        assert all(p.is_contiguous() for p in parts)


def test_two_rank_gloo_matches_single_process(tmp_path):
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0 = np.load(tmp_path / "rank0.npz")
    w1 = np.load(tmp_path / "rank1.npz")
    # single-process reference on the global batches, starting from rank 0's weights
    x, y = O.synth_columns(N_ROWS, seed=3)
    ws = O.glorot_init(CFG, seed=0)
    opt = O.Optimizer("Adam")
    perm = torch.randperm(N_ROWS, generator=torch.Generator().manual_seed(7)).numpy()
    for s in range(STEPS):
        idx = perm[s * GLOBAL_BATCH:(s + 1) * GLOBAL_BATCH]
        ws, _, _ = O.train_step(ws, opt, x[idx], y[idx], CFG, 1e-3)
    for i, ref in enumerate(ws):
        a, b = w0[f"arr_{i}"], w1[f"arr_{i}"]
        np.testing.assert_array_equal(a, b)                       # ranks stay bit-identical
        np.testing.assert_allclose(a, ref, rtol=2e-4, atol=2e-6)  # == training on the global batch


def test_two_rank_gloo_bf16_gradient_payload(tmp_path):
    """DataParallel(grad_payload="bf16") - what cs_dp_allreduce_bf16 does on the GPU, here through torch.distributed: the
    sums cross as bf16, both ranks receive the same buffer, training stays within bf16 gradient rounding of the fp32 run."""
    port = 29500 + (os.getpid() + 37) % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path), "bf16"), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "sum0.npy"), np.load(tmp_path / "sum1.npy")
    np.testing.assert_array_equal(s0, s1)
    n = s0.size
    parts = [(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) * 3.0).to(torch.bfloat16) for r in range(2)]
    want = (parts[0].float() + parts[1].float()).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(s0, want)
    w0 = np.load(tmp_path / "rank0.npz")
    w1 = np.load(tmp_path / "rank1.npz")
    x, y = O.synth_columns(N_ROWS, seed=3)
    ws = O.glorot_init(CFG, seed=0)
    opt = O.Optimizer("Adam")
    perm = torch.randperm(N_ROWS, generator=torch.Generator().manual_seed(7)).numpy()
    for s in range(STEPS):
        idx = perm[s * GLOBAL_BATCH:(s + 1) * GLOBAL_BATCH]
        ws, _, _ = O.train_step(ws, opt, x[idx], y[idx], CFG, 1e-3)
    for i, ref in enumerate(ws):
        a, b = w0[f"arr_{i}"], w1[f"arr_{i}"]
        np.testing.assert_array_equal(a, b)                       # replicas stay bit-identical to each other
        # Adam normalises the step: a 2^-8 relative change of a gradient moves a weight by a small fraction of lr per step
        np.testing.assert_allclose(a, ref, rtol=0, atol=STEPS * 1e-3 * 0.05)
