"""CPU ORACLE for the online-testing MLP -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/ may import this module; the product path (climsim_amd/online_mlp.py -> libclimsim_hip.so) never does.

What it restates (numpy, float32 with float64-accumulated products):
  * model : online_testing/baseline_models/MLP_v2rh/training/mlp.py:28-67 - `Linear -> ReLU` per hidden layer
            (dropout 0), `final_linear`, output pruning `x[:, 60:60+lev] = 0` (and 120:, 180:, 240:), ReLU on the last 8.
  * loss  : nn.MSELoss / nn.L1Loss / nn.SmoothL1Loss (beta 1), train_mlp_h5loader.py:226-236 (all column weights 1,
            :238-256 returns `criterion(pred, target)` then).
  * optimiser : torch.optim.Adam(lr), train_mlp_h5loader.py:210-211 - torch 2.x `_single_tensor_adam`
            (lerp for m, addcmul for v, denom = sqrt(v)/sqrt(1-b2^t) + eps, step lr/(1-b1^t), eps 1e-8).

PINNED: tests/test_online_mlp_cpu.py checks every function here against tests/golden/online_mlp_golden.npz, which
tests/golden/make_online_mlp_golden.py produced by running the reference's own MLP class, torch losses, autograd and
torch.optim.Adam in the build container.

`bf16=True` reproduces the HIP engine's rounding points (operands of every contraction rounded to bfloat16 RNE, fp32
accumulation, fp32 master weights), as in oracle/mlp_oracle.py.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .mlp_oracle import _mm, bf16_round

F32 = np.float32
Pairs = List[Tuple[np.ndarray, np.ndarray]]       # [(W (in,out), b)] hidden layers then the final linear


def from_state_dict(sd: Dict[str, np.ndarray]) -> Pairs:
    """torch state_dict (mlp.py:41-52: linears.{i}.0.weight (out,in), .bias, final_linear.*) -> [(W(in,out), b)]."""
    n_hidden = len([k for k in sd if k.startswith("linears.") and k.endswith(".weight")])
    keys = [f"linears.{i}.0" for i in range(n_hidden)] + ["final_linear"]
    return [(np.ascontiguousarray(np.asarray(sd[k + ".weight"], F32).T), np.asarray(sd[k + ".bias"], F32).copy()) for k in keys]


def to_state_dict(pairs: Pairs) -> Dict[str, np.ndarray]:
    sd = {}
    for i, (w, b) in enumerate(pairs):
        k = f"linears.{i}.0" if i + 1 < len(pairs) else "final_linear"
        sd[k + ".weight"], sd[k + ".bias"] = np.ascontiguousarray(w.T), b.copy()
    return sd


def keep_mask(n_out: int, output_prune: bool, strato_lev_out: int) -> np.ndarray:
    """1 for the columns MLP.forward produces, 0 for the ones it zeroes (mlp.py:56-61)."""
    keep = np.ones(n_out, F32)
    if output_prune:
        for start in (60, 120, 180, 240):
            keep[start:start + strato_lev_out] = 0
    return keep


def _lowbias32(x):
    x = np.asarray(x, np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D); x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B); x ^= x >> np.uint32(16)
    return x


def dropout_key_mlp(seed: int, step: int, layer: int) -> np.uint32:
    """Key of the mask of hidden layer `layer` (0-based) at optimiser step `step` (0-based: `iterations` before the step) -
    climsim_hip.hip chain_fwd_args."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    base = _lowbias32(np.uint32(seed & 0xFFFFFFFF) ^ _lowbias32(np.uint32(((seed >> 32) + 0x9E3779B9) & 0xFFFFFFFF)))
    return _lowbias32(np.uint32((int(base) + 0x9E3779B9 * (layer + 1) + 0x85EBCA6B * (step + 1)) & 0xFFFFFFFF))


def dropout_keep_mlp(seed: int, step: int, layer: int, rows: int, width: int, rate: float) -> np.ndarray:
    """Boolean keep mask (rows, width) of training-mode nn.Dropout(rate) as the engine draws it (kernels.h mlp_drop_hash2):
    one 32-bit hash per column pair of a row, 16 bits per element, keep iff bits >= floor(rate * 65536).  (torch's own
    Philox stream, mlp.py:44, cannot be reproduced outside torch: engine and oracle share this counter hash instead.)"""
    m = np.arange(rows, dtype=np.uint64)[:, None]
    n = np.arange(width, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        k = ((m * np.uint64(512) + (n >> np.uint64(1))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        k = k ^ ((m >> np.uint64(23)).astype(np.uint32) * np.uint32(0x9E3779B9))
    h = _lowbias32(k ^ dropout_key_mlp(seed, step, layer))
    bits = np.where((n & np.uint64(1)).astype(bool), h >> np.uint32(16), h & np.uint32(0xFFFF))
    return bits >= np.uint32(int(rate * 65536.0))


def forward(pairs: Pairs, x, keep, n_relu: int = 8, bf16: bool = False, keep_acts: bool = False, dropout=None):
    """`dropout` = None (eval mode) or (rate, seed, step): training-mode forward, relu(dropout(linear(x))) (mlp.py:41-52),
    kept activations scaled by float32 1/(1-rate)."""
    q = bf16_round if bf16 else (lambda a: np.asarray(a, F32))
    h = q(x)
    hs = [h]
    for li, (w, b) in enumerate(pairs[:-1]):
        a = np.maximum(_mm(h, q(w)) + b, 0).astype(F32)
        if dropout is not None and dropout[0] > 0:
            rate, seed, step = dropout
            scale = F32(1) / (F32(1) - F32(rate))
            a = np.where(dropout_keep_mlp(seed, step, li, a.shape[0], a.shape[1], rate), a * scale, F32(0)).astype(F32)
        h = q(a)
        hs.append(h)
    w, b = pairs[-1]
    y = (_mm(h, q(w)) + b).astype(F32)
    y = y * keep                                               # x[:, cols] = 0
    y[:, -n_relu:] = np.maximum(y[:, -n_relu:], 0)             # relu on the last 8
    return (y, hs) if keep_acts else y


def loss_value(pred, y, kind: str) -> float:
    e = pred.astype(np.float64) - np.asarray(y, np.float64)
    if kind == "mse":
        return float(np.mean(e ** 2))
    if kind == "mae":
        return float(np.mean(np.abs(e)))
    if kind == "huber":
        a = np.abs(e)
        return float(np.mean(np.where(a < 1, 0.5 * e ** 2, a - 0.5)))
    raise ValueError(kind)


def loss_and_grads(pairs: Pairs, x, y, keep, kind: str = "mse", n_relu: int = 8, bf16: bool = False, dropout=None):
    """(loss, [(dW (in,out), db)], prediction): gradients of the MEAN loss over batch x outputs.  `dropout` as in forward."""
    q = bf16_round if bf16 else (lambda a: np.asarray(a, F32))
    pred, hs = forward(pairs, x, keep, n_relu, bf16, keep_acts=True, dropout=dropout)
    bscale = F32(1) / (F32(1) - F32(dropout[0])) if dropout is not None and dropout[0] > 0 else F32(1)
    e = (pred - np.asarray(y, F32)).astype(F32)
    if kind == "mse":
        dz = 2 * e
    elif kind == "mae":
        dz = np.sign(e)
    elif kind == "huber":
        dz = np.clip(e, -1, 1)
    else:
        raise ValueError(kind)
    dz = dz.astype(F32) * keep
    dz[:, -n_relu:] *= (pred[:, -n_relu:] > 0)
    dz = q(dz)
    scale = F32(1.0 / pred.size)
    grads = [None] * len(pairs)
    for li in range(len(pairs) - 1, -1, -1):
        w, _ = pairs[li]
        grads[li] = ((_mm(hs[li].T, dz) * scale).astype(F32), (dz.astype(np.float64).sum(axis=0) * scale).astype(F32))
        if li > 0:
            dz = q((_mm(dz, q(w).T) * (hs[li] > 0) * bscale).astype(F32))       # a dropped unit has h = 0: no gradient
    return loss_value(pred, y, kind), grads, pred


class TorchAdam:
    """torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad), float32 state."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.t, self.m, self.v = 0, None, None

    def apply(self, pairs: Pairs, grads) -> Pairs:
        flat_p = [a for pr in pairs for a in pr]
        flat_g = [a for gr in grads for a in gr]
        if self.m is None:
            self.m = [np.zeros_like(a) for a in flat_p]
            self.v = [np.zeros_like(a) for a in flat_p]
        self.t += 1
        bc1 = 1.0 - self.beta1 ** self.t
        bc2_sqrt = np.sqrt(1.0 - self.beta2 ** self.t)
        step = F32(self.lr / bc1)
        out = []
        for i, (p, g) in enumerate(zip(flat_p, flat_g)):
            g = g.astype(F32)
            self.m[i] = (self.m[i] + F32(1 - self.beta1) * (g - self.m[i])).astype(F32)
            self.v[i] = (self.v[i] * F32(self.beta2) + (F32(1 - self.beta2) * g) * g).astype(F32)
            denom = (np.sqrt(self.v[i]) / F32(bc2_sqrt) + F32(self.eps)).astype(F32)
            out.append((p - (step * self.m[i]) / denom).astype(F32))
        return [(out[2 * i], out[2 * i + 1]) for i in range(len(pairs))]
