"""The data-parallel code path on the GPU with a one-rank RCCL process group (backend "nccl"): weight broadcast,
all-reduce of the flat gradient tensor, optimiser scaling by the GLOBAL batch.  With world_size 1 the result must equal
plain single-GPU training; the multi-rank arithmetic is covered on the CPU by tests/test_dp_gloo.py."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import mlp_oracle as O  # noqa: E402
from oracle import cnn_oracle as CO  # noqa: E402


@pytest.fixture(scope="module")
def pg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    from climsim_amd import build
    build.build()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_mlp_fit_distributed_equals_single(pg):
    from climsim_amd.mlp import MLPEmulator
    x, y = O.synth_columns(4096, seed=4)
    hist = []
    for distributed in (False, True):
        m = MLPEmulator(units=(128, 128), max_batch=512, seed=7)
        h = m.fit(x, y, batch_size=512, epochs=2, learning_rate=1e-3, seed=3, distributed=distributed)
        hist.append((h["loss"], m.get_weights()))
    np.testing.assert_allclose(hist[0][0], hist[1][0], rtol=1e-3)
    for a, b in zip(hist[0][1], hist[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4 * max(1.0, float(np.abs(a).max())))


def test_cnn_fit_distributed_runs(pg):
    from climsim_amd.cnn import CNNEmulator
    _, _, x3, y3 = CO.synth_cnn_columns(256, seed=11)
    m = CNNEmulator(depth=2, channel_width=64, max_batch=64, trainable=True, init_seed=1, seed=3)
    h = m.fit(x3, y3, batch_size=64, epochs=4, learning_rate=2e-3, distributed=True)
    assert h["loss"][-1] < 0.8 * h["loss"][0] and m.iterations == 16


def test_native_rccl_communicator_through_the_c_abi(pg):
    """cs_dp_* (include/climsim_hip.h): the engine's own RCCL communicator, collective issued on the compute stream.
    One rank: the sum over the group is the buffer itself; DataParallel takes this path on the GPU."""
    import ctypes as C
    from climsim_amd import _lib
    from climsim_amd.dp import DataParallel, RcclComm
    from climsim_amd.mlp import MLPEmulator
    comm = RcclComm(pg, torch.device("cuda", 0))
    t = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    ref = t.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # ordered with whatever stream is current, no events needed
        t.mul_(2.0)
        comm.all_reduce(t)
        t.mul_(0.5)
    side.synchronize()
    assert torch.equal(t, ref)
    lib = _lib.load()
    h = C.c_void_p()
    ident = C.create_string_buffer(128)
    assert lib.cs_dp_init(C.byref(h), None, ident, 2, 5, 0) != 0 and b"rank 5 of 2" in lib.cs_last_error()
    assert lib.cs_dp_allreduce(None, None, 0, None) != 0
    comm.close()
    m = MLPEmulator(units=(128,), max_batch=128, seed=1)
    dp = DataParallel(m, pg)
    assert dp.native is not None                        # the GPU path does not go through torch.distributed.all_reduce
    x, y = O.synth_columns(128, seed=2)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.loss_grads(xd, yd)
    g0 = m.gradient_tensor().clone()
    dp.all_reduce_grads()
    torch.cuda.synchronize()
    assert torch.equal(m.gradient_tensor(), g0)
    dp.close()


def test_bf16_gradient_payload_through_the_c_abi(pg):
    """cs_dp_allreduce_bf16: the buffer crosses the links as bf16 (round-to-nearest-even) and comes back widened; with one
    rank the result is exactly the bf16 rounding of the input, for every tail length of the 8-wide pack kernels."""
    from climsim_amd.dp import DataParallel, RcclComm
    from climsim_amd.mlp import MLPEmulator
    comm = RcclComm(pg, torch.device("cuda", 0))
    g = torch.Generator(device="cuda").manual_seed(5)
    for n in (1, 7, 8, 2048, 2049, 1_194_880 + 3):
        t = torch.randn(n + 4, device="cuda", generator=g)[:n] * 37.0          # 16-byte aligned start, ragged end
        t[::5] = 0.0
        want = t.to(torch.bfloat16).to(torch.float32)
        guard = t.clone()
        comm.all_reduce(t, "bf16")
        torch.cuda.synchronize()
        assert torch.equal(t, want), n
        assert not torch.equal(want, guard) or n == 1
    comm.close()
    m = MLPEmulator(units=(128,), max_batch=128, seed=1)
    dp = DataParallel(m, pg, grad_payload="bf16")
    assert dp.native is not None and dp.payload == "bf16"
    x, y = O.synth_columns(128, seed=2)
    m.loss_grads(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    g0 = m.gradient_tensor().clone()
    dp.all_reduce_grads()
    torch.cuda.synchronize()
    assert torch.equal(m.gradient_tensor(), g0.to(torch.bfloat16).to(torch.float32))
    dp.close()
    with pytest.raises(ValueError):
        DataParallel(m, pg, grad_payload="fp8")


def test_streamed_trainer_with_a_process_group(pg, tmp_path):
    """climsim_amd/stream.py under data parallelism (one rank): same weights as the single-process pass."""
    from climsim_amd.loader import GpuColumnLoader
    from climsim_amd.mlp import MLPEmulator
    from climsim_amd.stream import StreamedTrainer
    import types
    vin = ["state_t", "state_q0001", "state_ps", "pbuf_SOLIN", "pbuf_LHFLX", "pbuf_SHFLX"]
    vout = ["ptend_t", "ptend_q0001", "cam_out_NETSW", "cam_out_FLWDS", "cam_out_PRECSC", "cam_out_PRECC", "cam_out_SOLS", "cam_out_SOLL",
            "cam_out_SOLSD", "cam_out_SOLLD"]
    lens = {v: 60 if v in ("state_t", "state_q0001", "ptend_t", "ptend_q0001") else 1 for v in vin + vout}
    rng = np.random.default_rng(0)
    norm = (rng.normal(0, 1, 124), rng.uniform(0.5, 2, 124), rng.uniform(0.5, 2, 128))
    du = types.SimpleNamespace(input_vars=vin, target_vars=vout, var_lens=lens, normalize=True, input_abbrev="mli", output_abbrev="mlo",
                               save_norm=lambda: norm)
    ld = GpuColumnLoader(du)
    chunks = []
    for c in range(3):
        mli = rng.normal(0, 0.3, (2, 124, 200))
        mlo = rng.normal(0, 0.05, (2, 128, 200))
        mlo[:, :120] = mli[:, :120] + 1200.0 * mlo[:, :120]
        chunks.append((mli, mlo))
    out = []
    for dist in (None, pg):
        m = MLPEmulator(units=(128, 128), max_batch=128, seed=5)
        st = StreamedTrainer(m, ld, batch_size=128, dist=dist)
        res = st.fit_chunks(iter(chunks), learning_rate=1e-3, seed=3)
        out.append((res, m.get_weights()))
    assert out[0][0]["steps"] == out[1][0]["steps"] == -(-3 * 400 // 128)      # remainder rows are carried across chunks: one short batch per pass
    assert abs(out[0][0]["loss"] - out[1][0]["loss"]) <= 1e-3 * out[0][0]["loss"]
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-3 * max(1.0, float(np.abs(a).max())))
