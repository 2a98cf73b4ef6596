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
