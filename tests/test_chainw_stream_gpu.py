"""The wide chain's continuous weight stream (csrc/chainw.h: cws_tile, default) against its per-pass form (`CS_CHAINW_STREAM=0`): one
accumulator tile in k order either way, so losses, predictions, gradients and the weights after optimiser steps are BIT-IDENTICAL -
including shapes where a wave has no tile in a stage (128 wide), four tiles (1024 wide), ragged batches and the dropout epilogue.
(The reference side of the same kernels: test_hot_mlp_gpu.py holds the published widths at 3072 columns to torch's own vectors.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _with_stream(flag, make):
    old = os.environ.get("CS_CHAINW_STREAM")
    os.environ["CS_CHAINW_STREAM"] = flag
    try:
        return make()
    finally:
        if old is None:
            del os.environ["CS_CHAINW_STREAM"]
        else:
            os.environ["CS_CHAINW_STREAM"] = old


CASES = [
    ((768, 640, 512, 640, 640), "leakyrelu", "RAdam", 3072),      # the published model at its batch
    ((768, 640, 512, 640, 640), "relu", "Adam", 1000),            # ragged last row tile
    ((640, 128, 1024), "leakyrelu", "Adam", 2048),                # a stage with tiles on four waves only; four tiles per wave
    ((896, 384), "relu", "AdamTorch", 4100),
    ((256,) * 10, "relu", "Adam", 777),                           # twelve layers: the fetch cursor across many stages, two waves without tiles in every one
    ((1024, 128, 128, 1024), "leakyrelu", "Adam", 96),            # consecutive 128-wide stages (waves 4-7 skip two stages in a row), three row tiles
]


@pytest.mark.parametrize("units,act,opt,batch", CASES)
def test_stream_equals_the_per_pass_form(units, act, opt, batch):
    from climsim_amd.group import kernel_family
    from climsim_amd.mlp import MLPEmulator
    rng = np.random.default_rng(7)
    x = torch.from_numpy((rng.standard_normal((batch, 124)) * 0.3).astype(np.float32)).cuda()
    y = torch.from_numpy((rng.standard_normal((batch, 128)) * 0.05).astype(np.float32)).cuda()
    out = {}
    for flag in ("1", "0"):
        m = _with_stream(flag, lambda: MLPEmulator(units=units, activation=act, optimizer=opt, max_batch=batch, seed=3))
        assert kernel_family(m) == 2                              # the wide chain
        losses = [float(m.train_on_batch(x, y, 1e-3).cpu().numpy()[0]) for _ in range(3)]
        lg = float(m.loss_grads(x, y).cpu().numpy()[0])
        grads = m.get_gradients() if hasattr(m, "get_gradients") else None
        out[flag] = (losses, lg, grads, m.get_weights(), m.predict(x[:777].cpu().numpy()))
        m.close()
    a, b = out["1"], out["0"]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6)             # (loss sums: float atomics over the row tiles)
    assert a[1] == pytest.approx(b[1], rel=1e-6)
    for wa, wb in zip(a[3], b[3]):
        assert np.array_equal(wa, wb)
    assert np.array_equal(a[4], b[4])
    if a[2] is not None:
        for ga, gb in zip(a[2], b[2]):
            np.testing.assert_allclose(ga, gb, rtol=0, atol=1e-6 * max(1e-30, float(np.abs(gb).max())))     # (split-k atomics)


def test_stream_with_dropout_equals_the_per_pass_form():
    from climsim_amd import online_mlp as OM
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((900, 124)) * 0.3).astype(np.float32)
    y = (rng.standard_normal((900, 128)) * 0.05).astype(np.float32)
    out = {}
    for flag in ("1", "0"):
        m = _with_stream(flag, lambda: OM.MLP(124, 128, [384, 256, 640], 3, dropout=0.2, max_batch=1024, seed=5, dropout_seed=99))
        losses = [m.train_step(x, y, 1e-3) for _ in range(3)]
        out[flag] = (losses, m.state_dict(), m.forward(x, as_numpy=True))
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-6)
    for k in out["1"][1]:
        assert np.array_equal(out["1"][1][k], out["0"][1][k]), k
    assert np.array_equal(out["1"][2], out["0"][2])
