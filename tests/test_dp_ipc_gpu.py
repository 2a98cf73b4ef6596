"""One-shot all-reduce over peer-mapped buffers (cs_dp_ipc_*, climsim_amd/dp.py: IpcComm): TWO processes on
ONE GPU (the pool gives one GPU per box) exchange HIP IPC handles of their gradient buffers through the gloo rendezvous and
run the one-kernel reduce-scatter (pull) + all-gather (push).  Checked: (1) the raw collective against the exact float32 sum,
for several lengths incl. a ragged last slice, many steps in a row (the flag epochs); (2) a data-parallel `fit` with
collective="oneshot" against single-process training on the global batch - the same test test_dp_two_ranks_gpu.py runs with
gloo - with the replicas bit-identical to each other.  What this cannot show: behaviour over xGMI links between different
GPUs (cache coherence of remote lines); the protocol uses system-scope accesses on every peer word for that reason."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

UNITS, ROWS, GLOBAL_BATCH, EPOCHS = (128, 256), 4096, 512, 2


class _Engine:
    """The smallest thing DataParallel / IpcComm need: a rebindable flat float32 gradient tensor."""

    def __init__(self, n):
        self.g = torch.zeros(n, dtype=torch.float32, device="cuda")

    def gradient_tensor(self):
        return self.g

    def bind_gradient_tensor(self, t):
        self.g = t[:self.g.numel()]
        self.g.zero_()


def _raw_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from climsim_amd.dp import DataParallel
    ok = True
    for n in (4, 1000, 1196800, 262148):                       # 1196800 = the cfg-MLP's gradient; 262148 / 2 is not a multiple of 4 pieces
        eng = _Engine(n)
        dp = DataParallel(eng, dist, collective="oneshot")
        g = eng.gradient_tensor()
        for step in range(6):
            gen = torch.Generator(device="cuda").manual_seed(100 * step + rank)
            mine = torch.randn(n, device="cuda", generator=gen)
            other = torch.randn(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 * step + (1 - rank)))
            g.copy_(mine)
            dp.all_reduce_grads()
            torch.cuda.synchronize()
            want = (mine + other) if rank == 0 else (other + mine)      # the kernel adds in rank order: rank 0's value first
            ok = ok and bool(torch.equal(g, want)) and dp.native.timeouts == 0
        dp.close()
    np.save(os.path.join(out_dir, f"raw{rank}.npy"), np.asarray([ok]))
    dist.destroy_process_group()


def _fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", CS_DP_COLLECTIVE="oneshot")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from climsim_amd.mlp import MLPEmulator
    from oracle import mlp_oracle as O
    x, y = O.synth_columns(ROWS, seed=4)
    m = MLPEmulator(units=UNITS, max_batch=GLOBAL_BATCH, seed=7 + rank)      # ranks start DIFFERENT: the broadcast must fix it
    h = m.fit(x, y, batch_size=GLOBAL_BATCH, epochs=EPOCHS, learning_rate=1e-3, seed=3, distributed=True)
    np.savez(os.path.join(out_dir, f"fit{rank}.npz"), *m.get_weights(), loss=np.asarray(h["loss"]))
    dist.destroy_process_group()


def _spawn(fn, tmp_path):
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 90
    mp.spawn(fn, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_one_shot_allreduce_is_the_exact_sum(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    _spawn(_raw_worker, tmp_path)
    assert all(bool(np.load(tmp_path / f"raw{r}.npy")[0]) for r in range(2))


def test_data_parallel_fit_over_the_one_shot_collective_equals_single_rank_training(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd.mlp import MLPEmulator
    from oracle import mlp_oracle as O
    _spawn(_fit_worker, tmp_path)
    r0, r1 = np.load(tmp_path / "fit0.npz"), np.load(tmp_path / "fit1.npz")
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k])            # replicas bit-identical: same reduced buffer, same update
    x, y = O.synth_columns(ROWS, seed=4)
    m = MLPEmulator(units=UNITS, max_batch=GLOBAL_BATCH, seed=7)
    h = m.fit(x, y, batch_size=GLOBAL_BATCH, epochs=EPOCHS, learning_rate=1e-3, seed=3)
    np.testing.assert_allclose(r0["loss"], np.asarray(h["loss"]), rtol=2e-3)
    for i, w in enumerate(m.get_weights()):
        assert np.linalg.norm(r0[f"arr_{i}"] - w) <= 2e-2 * np.linalg.norm(w) + 1e-6
