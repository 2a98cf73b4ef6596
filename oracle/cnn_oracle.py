"""CPU ORACLE for the level-axis CNN -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `CNNHyperModel.build` (baseline_models/CNN/training/hpo_train.py:124-200) with torch-CPU
float32 ops: hp_depth blocks of [Conv1D(C,3,'same') -> ReLU -> Dropout] x2 plus a Conv1D(C,1)
projection of the block input added after the second activation; Conv1D(10,1, ELU); per-level
Dense(10->2, linear) || Dense(10->8, relu); losses `mae_adjusted` / `mse_adjusted` (:114-121).
Conv1D/Dense semantics come from un-vendored Keras (TF 2.10, CNN/env/tf2.yml:205-223): kernels
(k, c_in, c_out), zero 'same' padding, glorot_uniform, zero biases.  PARITY UNPINNED: the reference
holds no test or golden vector for the CNN (saved_model.pb has no variables); the known answer that
exists - 13,215,420 parameters for depth 12 / width 406 - is asserted in tests.

`bf16=True` mirrors the engine's rounding points: weights, and every stored activation tensor
(conv outputs after activation, the residual projection, the block sum) rounded to bfloat16.
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
        r = _q(_conv(x, wr, br, bf16), bf16)
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
