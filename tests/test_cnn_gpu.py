"""GPU parity of the CNN prediction path against the torch-CPU oracle (bf16-emulating and fp32).
Tolerances: bf16-emulating oracle max|d| <= 5e-3*max|ref| (accumulation order + rare bf16 flips through
37 conv layers); fp32 oracle <= 6e-2*max|ref| (bf16 operand rounding)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import cnn_oracle as CO  # noqa: E402
from golden_inputs import det_uniform  # noqa: E402


@pytest.fixture(scope="module")
def CNN():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd import cnn
    return cnn


def make_input(n, seed):
    x = (det_uniform(n * 124, seed).reshape(n, 124) - 0.5).astype(np.float32)
    return x


@pytest.mark.parametrize("depth,width,n", [(2, 128, 5), (3, 406, 37), (12, 406, 16)])
def test_cnn_forward_matches_oracle(CNN, depth, width, n):
    from climsim_amd.data_utils import data_utils
    ws = CO.glorot_cnn(seed=depth, bias_scale=0.05, depth=depth, channels=width)
    m = CNN.CNNEmulator(depth=depth, channel_width=width, max_batch=16)
    assert m.count_params() == sum(w.size for w in ws)
    m.set_weights(ws)
    x = make_input(n, 40 + depth)
    x3 = data_utils.reshape_input_for_cnn(x)
    got_flat_in = m.predict(x)                       # (N,124) rows: reshape happens on the GPU
    got_3d_in = m.predict(x3)                        # materialised (N,60,6)
    np.testing.assert_array_equal(got_flat_in, got_3d_in)
    ref16 = CO.forward(ws, x3, depth=depth, bf16=True)
    ref32 = CO.forward(ws, x3, depth=depth, bf16=False)
    assert got_3d_in.shape == (n, 60, 10)
    assert np.max(np.abs(got_3d_in - ref16)) <= 5e-3 * np.max(np.abs(ref16))
    assert np.max(np.abs(got_3d_in - ref32)) <= 6e-2 * np.max(np.abs(ref32))
    assert np.all(got_3d_in[:, :, 2:] >= 0)
    flat = m.predict(x, flat_output=True)
    np.testing.assert_allclose(flat, data_utils.reshape_target_from_cnn(got_3d_in), rtol=1e-5, atol=1e-6)


def test_cnn_param_count_published_model(CNN):
    m = CNN.CNNEmulator(depth=12, channel_width=406, max_batch=4)
    assert m.count_params() == 13_215_420            # BASELINE.md (hpo_train.py:137-200)
    with pytest.raises(ValueError):
        m.set_weights([np.zeros((3, 6, 406), np.float32)])
    with pytest.raises(ValueError):
        m.predict(np.zeros((2, 100), np.float32))
