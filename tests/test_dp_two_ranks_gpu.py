"""Two data-parallel ranks with the REAL HIP engine on ONE GPU (the pool gives one GPU per box): both processes run on
cuda:0, the collective is torch.distributed's gloo all-reduce on device tensors (CS_DP_NATIVE=0; RCCL cannot put two
ranks on one device).  What is under test is everything around the collective that the driver's 8-GPU run depends on:
round-robin sharding of the global batch, UNSCALED gradient sums, 1/(128*global_batch) in the optimiser kernel, weight
broadcast, per-epoch loss reduction - against single-rank training on the global batch (the pattern of
online_testing/baseline_models/MLP_v2rh/training/train_mlp_h5loader.py:195-207, 470-473)."""
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


def _data():
    from oracle import mlp_oracle as O
    return O.synth_columns(ROWS, seed=4)


def _fit_worker(rank, world, port, out_dir, payload="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CS_DP_NATIVE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      CS_DP_PAYLOAD=payload)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from climsim_amd.mlp import MLPEmulator
    x, y = _data()
    m = MLPEmulator(units=UNITS, max_batch=GLOBAL_BATCH, seed=7 + rank)      # ranks start DIFFERENT: the broadcast must fix it
    h = m.fit(x, y, batch_size=GLOBAL_BATCH, epochs=EPOCHS, learning_rate=1e-3, seed=3, distributed=True)
    np.savez(os.path.join(out_dir, f"fit{rank}.npz"), *m.get_weights(), loss=np.asarray(h["loss"]))
    dist.destroy_process_group()


def _stream_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CS_DP_NATIVE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import types
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from climsim_amd.loader import GpuColumnLoader
    from climsim_amd.mlp import MLPEmulator
    from climsim_amd.stream import StreamedTrainer
    vin = ["state_t", "state_q0001", "state_ps", "pbuf_SOLIN", "pbuf_LHFLX", "pbuf_SHFLX"]
    vout = ["ptend_t", "ptend_q0001", "cam_out_NETSW", "cam_out_FLWDS", "cam_out_PRECSC", "cam_out_PRECC", "cam_out_SOLS", "cam_out_SOLL",
            "cam_out_SOLSD", "cam_out_SOLLD"]
    lens = {v: 60 if v in ("state_t", "state_q0001", "ptend_t", "ptend_q0001") else 1 for v in vin + vout}
    rng0 = np.random.default_rng(0)
    norm = (rng0.normal(0, 1, 124), rng0.uniform(0.5, 2, 124), rng0.uniform(0.5, 2, 128))
    du = types.SimpleNamespace(input_vars=vin, target_vars=vout, var_lens=lens, normalize=True, input_abbrev="mli", output_abbrev="mlo",
                               save_norm=lambda: norm)
    ld = GpuColumnLoader(du)
    rng = np.random.default_rng(10 + rank)
    chunks = []
    # UNEQUAL streams: rank 1's chunks are wider (ragged against the batch) and it has one chunk more
    for c in range(3 + rank):
        ncol = 200 + 37 * rank
        mli = rng.normal(0, 0.3, (2, 124, ncol))
        mlo = rng.normal(0, 0.05, (2, 128, ncol))
        mlo[:, :120] = mli[:, :120] + 1200.0 * mlo[:, :120]
        chunks.append((torch.from_numpy(mli).cuda(), torch.from_numpy(mlo).cuda()) if c % 2 else (mli, mlo))
    m = MLPEmulator(units=(128, 128), max_batch=128, seed=5)
    st = StreamedTrainer(m, ld, batch_size=128, dist=dist)
    res = st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=3)
    np.savez(os.path.join(out_dir, f"stream{rank}.npz"), *m.get_weights(), steps=res["steps"], rows=res["rows"], dropped=st.rows_dropped)
    dist.destroy_process_group()


def _spawn(fn, tmp_path, *extra):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() + 17 * len(extra)) % 300
    mp.spawn(fn, args=(2, port, str(tmp_path), *extra), nprocs=2, join=True)


def test_two_ranks_on_one_gpu_equal_single_rank_training(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    _spawn(_fit_worker, tmp_path)
    w0, w1 = np.load(tmp_path / "fit0.npz"), np.load(tmp_path / "fit1.npz")
    from climsim_amd.mlp import MLPEmulator
    x, y = _data()
    m = MLPEmulator(units=UNITS, max_batch=GLOBAL_BATCH, seed=7)              # rank 0's initial weights
    h = m.fit(x, y, batch_size=GLOBAL_BATCH, epochs=EPOCHS, learning_rate=1e-3, seed=3)
    ref = m.get_weights()
    for i, r in enumerate(ref):
        a, b = w0[f"arr_{i}"], w1[f"arr_{i}"]
        np.testing.assert_array_equal(a, b)                                   # same reduced buffer, same update: bit-identical ranks
        np.testing.assert_allclose(a, r, rtol=0, atol=2e-4 * max(1.0, float(np.abs(r).max())))   # float atomics order only
    np.testing.assert_allclose(w0["loss"], h["loss"], rtol=1e-3)
    np.testing.assert_allclose(w1["loss"], h["loss"], rtol=1e-3)


def test_two_ranks_bf16_gradient_payload(tmp_path):
    """CS_DP_PAYLOAD=bf16 (DataParallel(grad_payload="bf16")): the gradient sums cross as bf16; the replicas still end
    bit-identical to each other and within bf16 gradient rounding of single-rank fp32 training (Adam normalises the step)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _spawn(_fit_worker, tmp_path, "bf16")
    w0, w1 = np.load(tmp_path / "fit0.npz"), np.load(tmp_path / "fit1.npz")
    from climsim_amd.mlp import MLPEmulator
    x, y = _data()
    m = MLPEmulator(units=UNITS, max_batch=GLOBAL_BATCH, seed=7)
    h = m.fit(x, y, batch_size=GLOBAL_BATCH, epochs=EPOCHS, learning_rate=1e-3, seed=3)
    steps = EPOCHS * (ROWS // GLOBAL_BATCH)
    moved = 0.0
    for i, r in enumerate(m.get_weights()):
        a, b = w0[f"arr_{i}"], w1[f"arr_{i}"]
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, r, rtol=0, atol=steps * 1e-3 * 0.05)
        moved = max(moved, float(np.abs(a - r).max()))
    assert moved > 0.0                                                        # the payload really was rounded
    np.testing.assert_allclose(w0["loss"], h["loss"], rtol=5e-3)


def test_two_ranks_streaming_unequal_chunks_stay_in_step(tmp_path):
    """Ranks that stream different amounts of data (wider chunks, one chunk more) must still issue the same collectives:
    per chunk they agree on the smallest row count, the pass ends when the first rank runs dry (climsim_amd/stream.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _spawn(_stream_worker, tmp_path)
    s0, s1 = np.load(tmp_path / "stream0.npz"), np.load(tmp_path / "stream1.npz")
    assert int(s0["steps"]) == int(s1["steps"]) == -(-3 * 400 // 128)         # 3 common chunks of 400 agreed rows, remainders carried: ceil(1200 / 128) batches
    assert int(s0["rows"]) == int(s1["rows"]) == 3 * 400
    assert int(s0["dropped"]) == 0 and int(s1["dropped"]) == 3 * 2 * 37       # rank 1's surplus columns
    for k in s0.files:
        if k.startswith("arr_"):
            np.testing.assert_array_equal(s0[k], s1[k])
