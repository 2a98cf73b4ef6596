"""torch-CPU float32 port of the reference MLP train step -- TEST INFRASTRUCTURE / CPU BASELINE.

Second, independent restatement of step2_retrain.py:95-162 (model, mse loss) that lets torch
autograd derive the backward pass, with a hand-written Keras-2.11 Adam (epsilon OUTSIDE the bias
correction, eps=1e-7; torch.optim.Adam differs).  Used (a) in tests/test_oracle.py to cross-check
the hand-derived gradients of oracle/mlp_oracle.py, (b) by bench.py's `cpu_baseline` leg as the
multi-threaded CPU "port" of the reference training path (TensorFlow/Keras are not installable
here).  Never imported by climsim_amd/.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .mlp_oracle import MLPConfig, fuse_heads


class _RoundSTE(torch.autograd.Function):
    """bf16 rounding of a tensor, gradient passed through: fp32 master weights / activations, bf16 MFMA operands."""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """identity forward, bf16 rounding of the gradient: where the engine stores dz as bf16 (operand of dgrad and wgrad)."""

    @staticmethod
    def forward(ctx, t):
        return t

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class TorchMLP:
    """`bf16=True`: the ENGINE's arithmetic on the CPU - every contraction's operands rounded to bfloat16 (inputs, activations, the
    weights' operand copies, dz), float32 accumulation, float32 master weights and optimiser - the rounding points of
    oracle/mlp_oracle.py's `bf16=True` (which tests hold the engine to step by step), fast enough to TRAIN with: bench.py's
    acceptance leg uses it to tell what bf16 operands cost after equal steps from what the engine would add on top."""

    def __init__(self, ws, cfg: MLPConfig, bf16: bool = False):
        self.cfg = cfg
        self.bf16 = bool(bf16)
        self.params = [torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
                       for pair in fuse_heads(ws) for a in pair]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.it = 0

    def forward(self, x):
        cfg, h = self.cfg, x
        n = len(self.params) // 2
        q = _RoundSTE.apply if self.bf16 else (lambda t: t)
        h = q(h)
        for i in range(n):
            z = torch.addmm(self.params[2 * i + 1], h, q(self.params[2 * i]))
            if self.bf16:
                z = _RoundGrad.apply(z)
            if i < n - 1:
                if cfg.act == "relu":
                    h = torch.relu(z)
                elif cfg.act == "elu":
                    h = torch.nn.functional.elu(z)
                else:
                    h = torch.nn.functional.leaky_relu(z, cfg.alpha)
                h = q(h)
            else:
                h = torch.cat([z[:, :cfg.n_out_lin], torch.relu(z[:, cfg.n_out_lin:])], dim=1)
        return h

    def loss_and_grads(self, x, y):
        for p in self.params:
            p.grad = None
        loss = torch.mean((self.forward(x) - y) ** 2)
        loss.backward()
        return float(loss.detach()), [p.grad for p in self.params]

    @torch.no_grad()
    def adam(self, grads, lr, b1=0.9, b2=0.999, eps=1e-7):
        self.it += 1
        alpha = lr * (1 - b2 ** self.it) ** 0.5 / (1 - b1 ** self.it)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.add_(g - m, alpha=1 - b1)
            v.add_(g * g - v, alpha=1 - b2)
            p.addcdiv_(m, v.sqrt().add_(eps), value=-alpha)

    def train_step(self, x, y, lr):
        loss, grads = self.loss_and_grads(x, y)
        self.adam(grads, lr)
        return loss


def _time_steps(model, xt, yt, batch, budget_s, warmup, min_steps):
    n = (xt.shape[0] // batch) * batch
    times, i, t_start = [], 0, time.perf_counter()
    while True:
        lo = (i * batch) % n
        t0 = time.perf_counter()
        model.train_step(xt[lo:lo + batch], yt[lo:lo + batch], 1e-3)
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
        i += 1
        if time.perf_counter() - t_start > budget_s and len(times) >= min_steps:
            return times


def time_cpu_baseline(ws, cfg, x, y, batch=1024, budget_s=15.0, warmup=5, threads=None):
    """Median columns/s of the CPU port on a bounded sample.  The thread count is chosen by a short
    trial over {all cores, 64, 32, 16, 8} (small GEMMs do not scale to 128 threads); `cores` in the
    result is the count actually used for the reported number."""
    ncpu = os.cpu_count() or 1
    xt, yt = torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(y))
    cands = [threads] if threads else sorted({c for c in (ncpu, 64, 32, 16, 8) if c <= ncpu}, reverse=True)
    best = (None, float("inf"))
    trial = min(2.0, budget_s / (2 * len(cands)))
    for c in cands:
        torch.set_num_threads(c)
        med = float(np.median(_time_steps(TorchMLP(ws, cfg), xt, yt, batch, trial, 2, 3)))
        if med < best[1]:
            best = (c, med)
    torch.set_num_threads(best[0])
    times = _time_steps(TorchMLP(ws, cfg), xt, yt, batch, budget_s / 2, warmup, 10)
    med = float(np.median(times))
    return {"value": batch / med, "unit": "columns/s", "cores": best[0], "kind": "port",
            "sample": f"{len(times)} steps of batch {batch}, fp32 torch-CPU (oneDNN), {best[0]} of {ncpu} threads "
                      f"(best of {cands}), median step {med * 1e3:.2f} ms"}
