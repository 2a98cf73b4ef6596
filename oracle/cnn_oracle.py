"""CPU ORACLE for the level-axis CNN -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `CNNHyperModel.build` (baseline_models/CNN/training/hpo_train.py:124-200) with torch-CPU
float32 ops: hp_depth blocks of [Conv1D(C,3,'same') -> ReLU -> Dropout] x2 plus a Conv1D(C,1)
projection of the block input added after the second activation; Conv1D(10,1, ELU); per-level
Dense(10->2, linear) || Dense(10->8, relu); losses `mae_adjusted` / `mse_adjusted` (:114-121).
Conv1D/Dense semantics come from un-vendored Keras (TF 2.10, CNN/env/tf2.yml:205-223): kernels
(k, c_in, c_out), zero 'same' padding, glorot_uniform, zero biases.  PARITY: the model's forward / backward is UNPINNED - the
reference holds no test or golden vector for the CNN (saved_model.pb has no variables) and TensorFlow cannot run here; the known
answer that exists - 13,215,420 parameters for depth 12 / width 406 - is asserted in tests.  PINNED (round 6): the
loss / metric FUNCTIONS `mae_adjusted`, `mse_adjusted`, `continuous_ranked_probability_score` - tests/golden/cnn_loss_golden.npz is made
by executing the reference's own three function bodies (hpo_train.py:83-121, taken out of the file by ast) over numpy namesakes of the
backend operations they call (tests/golden/make_cnn_loss_golden.py); tests/test_oracle.py holds the restatements below to it at 1e-12.

`bf16=True` mirrors the engine's rounding points: weights, and every stored activation tensor
(conv outputs after activation, the block sum) rounded to bfloat16; the residual projection is accumulated
in fp32 on top of the activated second conv inside one kernel and is never stored.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .mlp_oracle import bf16_round


def cnn_shapes(depth=12, channels=406, c_in=6, c_out=10, n_lin=2, k=3):
    """Keras weight order (layers in creation order): per block conv_a, conv_b, conv_residual."""
    shapes = []
    for b in range(depth):
        cin = c_in if b == 0 else channels
        shapes += [(k, cin, channels), (channels,), (k, channels, channels), (channels,), (1, cin, channels), (channels,)]
    shapes += [(1, channels, c_out), (c_out,), (c_out, n_lin), (n_lin,), (c_out, c_out - n_lin), (c_out - n_lin,)]
    return shapes


def glorot_cnn(seed=0, bias_scale=0.0, gain=1.0, **kw):
    rng = np.random.default_rng(seed)
    ws = []
    for s in cnn_shapes(**kw):
        if len(s) == 1:
            ws.append(rng.normal(0, bias_scale, s).astype(np.float32) if bias_scale else np.zeros(s, np.float32))
        else:
            rf = int(np.prod(s[:-2])) if len(s) == 3 else 1
            lim = np.sqrt(6.0 / (rf * s[-2] + rf * s[-1]))
            ws.append((rng.uniform(-lim, lim, s) * gain).astype(np.float32))
    return ws


def _q(t, bf16):
    return torch.from_numpy(bf16_round(t.detach().numpy())) if bf16 else t


def _conv(x, w, b, bf16):
    """x (B, L, Cin) channels-last; w Keras (k, Cin, Cout)."""
    wt = _q(torch.from_numpy(w), bf16).permute(2, 1, 0).contiguous()      # (Cout, Cin, k)
    y = F.conv1d(x.permute(0, 2, 1).double(), wt.double(), None, padding=w.shape[0] // 2).float()
    return y.permute(0, 2, 1) + torch.from_numpy(b)


def forward(ws, x3, depth=12, n_lin=2, bf16=False):
    """x3: (B, 60, 6) float32 -> (B, 60, 10) float32 (inference: dropout is identity)."""
    x = _q(torch.from_numpy(np.ascontiguousarray(x3, np.float32)), bf16)
    i = 0
    for _ in range(depth):
        wa, ba, wb, bb, wr, br = ws[i:i + 6]
        i += 6
        a1 = _q(torch.relu(_conv(x, wa, ba, bf16)), bf16)
        r = _conv(x, wr, br, bf16)                      # accumulated in fp32 on top of the activated conv b (same launch)
        x = _q(torch.relu(_conv(a1, wb, bb, bf16)) + r, bf16)
    wo, bo, wl, bl, wrel, brel = ws[i:i + 6]
    o = _q(F.elu(_conv(x, wo, bo, bf16)), bf16)
    lin = o @ torch.from_numpy(wl) + torch.from_numpy(bl)
    rel = torch.relu(o @ torch.from_numpy(wrel) + torch.from_numpy(brel))
    return torch.cat([lin, rel], dim=-1).numpy()


def mae_adjusted(y_true, y_pred):
    ae = np.abs(y_pred - y_true)
    return float(ae[:, :, 0:2].mean() * (120 / 128) + ae[:, :, 2:10].mean() * (8 / 128))


def mse_adjusted(y_true, y_pred):
    se = (y_pred - y_true) ** 2
    return float(se[:, :, 0:2].mean() * (120 / 128) + se[:, :, 2:10].mean() * (8 / 128))


def continuous_ranked_probability_score(y_true, y_pred):
    """hpo_train.py:83-111 restated in numpy: per (column, level) mean_j |p_j - y_j| - 1/2 mean_{i,j} |p_i - p_j| over the last axis
    (the 10 channels play the role of the forecast ensemble there), then the mean over everything else.  float64."""
    yt, yp = np.asarray(y_true, np.float64), np.asarray(y_pred, np.float64)
    score = np.abs(yp - yt).mean(axis=-1)
    diff = yp[..., :, None] - yp[..., None, :]
    score = score - 0.5 * np.abs(diff).mean(axis=(-2, -1))
    return float(score.mean())


def categorical_accuracy(y_true, y_pred):
    """metrics=["accuracy"] on (B, 60, 10) float targets resolves to Keras' categorical accuracy (hpo_train.py:231; keras 2.10
    compile_utils: last-axis size > 1, dense targets): argmax over the channels agrees (np.argmax = first index on ties, as tf.argmax)."""
    return float((np.argmax(np.asarray(y_true), axis=-1) == np.argmax(np.asarray(y_pred), axis=-1)).mean())


# ------------------------------------------------------------------------------------------------
# Training path (hpo_train.py:159-236 model + :114-121 losses; Keras autodiff restated with torch
# autograd on CPU).  PARITY UNPINNED like the forward: Keras' Dropout draws from TF's own random
# stream, which no other implementation can reproduce; the engine and this oracle share a
# counter-based hash instead (same Bernoulli(keep = 1-rate) law, same 1/keep scaling as
# keras.layers.Dropout).
# ------------------------------------------------------------------------------------------------
ROW_PITCH = 512          # channel pitch of the engine's activation rows (element id = row*512 + channel)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def dropout_key(seed, layer):
    """Per (step seed, dropout layer) key; layer = 2*block + {0: after conv a, 1: after conv b}."""
    with np.errstate(over="ignore"):
        return lowbias32(np.uint32(seed & 0xFFFFFFFF) + np.uint32(0x9E3779B9) * np.uint32(layer + 1))


def dropout_keep(seed, layer, m_rows, channels, rate):
    """Boolean keep mask (m_rows, channels).  One 32-bit hash per channel pair (n, n+1) of a row, 16 bits per
    element: keep iff its 16 bits >= floor(rate * 65536)."""
    m = np.arange(m_rows, dtype=np.uint64)[:, None]
    n = np.arange(channels, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        k = ((m * np.uint64(ROW_PITCH // 2) + (n >> np.uint64(1))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        k ^= ((m >> np.uint64(24)).astype(np.uint32) * np.uint32(0x9E3779B9))
        h = lowbias32(k ^ dropout_key(seed, layer))
    bits = np.where((n & np.uint64(1)).astype(bool), h >> np.uint32(16), h & np.uint32(0xFFFF))
    return bits >= np.uint32(int(rate * (1 << 16)))


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bfloat16 (a stored bf16 gradient tensor of the engine)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(bf16_round(g.numpy()))


def _st(t, bf16):
    """Straight-through bf16 rounding of a stored activation."""
    return t + (_q(t, True) - t).detach() if bf16 else t


def _rg(t, bf16):
    return _RoundGrad.apply(t) if bf16 else t


def _tconv(x, w, b, bf16):
    wq = _st(w, bf16).permute(2, 1, 0)
    return F.conv1d(x.permute(0, 2, 1), wq, None, padding=w.shape[0] // 2).permute(0, 2, 1) + b


def loss_and_grads(ws, x3, y3, depth=12, n_lin=2, loss="mae", rate=0.0, seed=0, bf16=False):
    """One training-mode pass.  x3 (B,60,6), y3 (B,60,10) float32.
    Returns (dict of loss values, list of gradients of the chosen loss in Keras weight order)."""
    B, L, _ = x3.shape
    P = [torch.from_numpy(np.ascontiguousarray(w, np.float32)).requires_grad_(True) for w in ws]
    x = _q(torch.from_numpy(np.ascontiguousarray(x3, np.float32)), bf16)
    keep_p = np.float32(1.0) - np.float32(rate)

    def drop(t, layer):
        if rate <= 0:
            return t
        C = t.shape[-1]
        k = torch.from_numpy(dropout_keep(seed, layer, B * L, C, rate).reshape(B, L, C))
        return torch.where(k, t * torch.tensor(np.float32(1.0) / keep_p), torch.zeros((), dtype=t.dtype))

    i = 0
    for blk in range(depth):
        wa, ba, wb, bb, wr, br = P[i:i + 6]
        i += 6
        za = _rg(_tconv(x, wa, ba, bf16), bf16)                       # dz1 is stored bf16
        a1 = _st(drop(torch.relu(za), 2 * blk), bf16)
        r = _rg(_tconv(x, wr, br, bf16), bf16)                        # projection branch sees the stored bf16 g
        zb = _rg(_tconv(a1, wb, bb, bf16), bf16)                      # dz2 = round(g_fp32 * mask / keep)
        x = _st(drop(torch.relu(zb), 2 * blk + 1) + r, bf16)
    wo, bo, wl, bl, wrel, brel = P[i:i + 6]
    zo = _rg(_tconv(x, wo, bo, bf16), bf16)
    o = _st(F.elu(zo), bf16)
    pred = torch.cat([o @ wl + bl, torch.relu(o @ wrel + brel)], dim=-1)
    e = pred - torch.from_numpy(np.ascontiguousarray(y3, np.float32))
    wp, ws_ = 120.0 / 128.0, 8.0 / 128.0
    mae = e[:, :, :n_lin].abs().mean() * wp + e[:, :, n_lin:].abs().mean() * ws_
    mse = (e[:, :, :n_lin] ** 2).mean() * wp + (e[:, :, n_lin:] ** 2).mean() * ws_
    (mae if loss == "mae" else mse).backward()
    out = {"mae_adjusted": float(mae.detach()), "mse_adjusted": float(mse.detach()), "pred": pred.detach().numpy()}
    return out, [p.grad.numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32) for p in P]


def synth_cnn_columns(n, seed=7):
    """Flat low-res-shaped columns (see mlp_oracle.synth_columns) in the (n,60,6)/(n,60,10) CNN layout of
    data_utils.reshape_input_for_cnn / reshape_target_for_cnn."""
    from .mlp_oracle import synth_columns
    x, y = synth_columns(n, seed=seed)
    x3 = np.concatenate([x[:, 0:60, None], x[:, 60:120, None], np.repeat(x[:, None, 120:124], 60, axis=1)], axis=2)
    y3 = np.concatenate([y[:, 0:60, None], y[:, 60:120, None], np.repeat(y[:, None, 120:128], 60, axis=1)], axis=2)
    return x, y, np.ascontiguousarray(x3, np.float32), np.ascontiguousarray(y3, np.float32)
