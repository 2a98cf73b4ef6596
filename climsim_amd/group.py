"""K models, one grouped launch per kernel kind (C ABI: cs_mlp_group_*, include/climsim_hip.h).

The reference's hyper-parameter search runs several worker processes per GPU, each training a small MLP at batch
48..3072 (baseline_models/MLP/training/HPO/baseline_v1/hpo_baseline_v1.py:221-245, 255-260), and its RPN baseline trains a
32-member ensemble of one shape (baseline_models/RPN/training/rpn_model_v1_data.py:71-163).  One such step occupies n/32 of
the 256 compute units and every workgroup streams all weights whatever n is, so the step time is flat from 1024 to 8192
columns.  `MLPGroup` steps K `MLPEmulator`s together: the layer chains of all members are ONE launch, their weight
gradients one, their optimisers one.  Members keep their own weights, optimiser rule and state, learning rate, batch and
checkpoints; per member the arithmetic is the one of `MLPEmulator.train_on_batch` (same kernel bodies).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from . import _lib


def kernel_family(model) -> int:
    """1 = tuned layer chain (hidden widths 128/256/512, 128 outputs), 2 = wide layer chain, 0 = one GEMM per layer
    (cannot be grouped); +16 for ELU.  Members of a group must agree."""
    return int(model.lib.cs_mlp_kernel_family(model._h))


class MLPGroup:
    def __init__(self, models: Sequence):
        import torch
        if not models:
            raise ValueError("a group needs at least one model")
        self.models = list(models)
        self.lib = self.models[0].lib
        self.device = self.models[0].device
        k = len(self.models)
        arr = (C.c_void_p * k)(*[m._h for m in self.models])
        self._g = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cs_mlp_group_create(C.byref(self._g), arr, k))
        self.k = k
        self._loss = torch.zeros((k, 2), dtype=torch.float32, device=self.device)

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self.lib.cs_mlp_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _args(self, x, y, lrs, row_idx, n, active):
        k = self.k

        def per_member(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * k
        xs, ys, idx = per_member(x), per_member(y), per_member(row_idx)
        lrs = [float(v) for v in (lrs if isinstance(lrs, (list, tuple)) else [lrs] * k)]
        act = [True] * k if active is None else [bool(a) for a in active]
        ns = []
        for i in range(k):
            if not act[i]:
                ns.append(0)
            elif n is not None:
                ns.append(int(n[i] if isinstance(n, (list, tuple)) else n))
            else:
                ns.append(int(idx[i].numel() if idx[i] is not None else xs[i].shape[0]))
        P = C.c_void_p
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        ax = (P * k)(*[ptr(t) for t in xs])
        ay = (P * k)(*[ptr(t) for t in ys])
        ai = (P * k)(*[ptr(t) for t in idx])
        an = (C.c_int64 * k)(*ns)
        al = (C.c_float * k)(*lrs)
        return ax, ay, ai, an, al, act

    def train_on_batch(self, x, y, lrs, row_idx=None, n=None, normalise: bool = False, loss=None, active=None):
        """One optimiser step of every (active) member.  `x`, `y`, `row_idx`: one tensor shared by all members or a list
        with one entry per member (float32 device rows / int64 row indices); `lrs`: float or list.  Returns the (k, 2)
        device tensor of [sum sq err, sum abs err] per member (rows of inactive members are left untouched)."""
        import torch
        loss = self._loss if loss is None else loss
        ax, ay, ai, an, al, act = self._args(x, y, lrs, row_idx, n, active)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.cs_mlp_group_train_step(self._g, ax, ay, ai, an, int(normalise), al, C.c_void_p(loss.data_ptr()), st))
        for m, a in zip(self.models, act):
            if a:
                m.iterations += 1
        return loss

    def forward_batch(self, x, yhat=None, y=None, row_idx=None, n=None, normalise: bool = False, loss=None, accumulate: bool = False,
                      active=None):
        """Prediction / evaluation of every (active) member on one batch in ONE launch (cs_mlp_group_forward).  `x`, `y`,
        `row_idx`, `yhat`: one tensor for all members or a list with one entry per member (a shared `yhat` tensor would be
        written by every member: pass a list, or a (k, n, n_out) tensor).  With `y`, the (k, 2) tensor `loss` receives
        [sum sq err, sum abs err] per member (added to when `accumulate`)."""
        import torch
        k = self.k
        if yhat is not None and not isinstance(yhat, (list, tuple)):
            if yhat.ndim != 3 or yhat.shape[0] != k:
                raise ValueError("yhat must be a list of k tensors or one (k, n, n_out) tensor")
            yhat = [yhat[i] for i in range(k)]
        loss = self._loss if (loss is None and y is not None) else loss
        ax, ay, ai, an, _, act = self._args(x, y, 0.0, row_idx, n, active)
        P = C.c_void_p
        ah = (P * k)(*[t.data_ptr() if t is not None else None for t in yhat]) if yhat is not None else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.cs_mlp_group_forward(self._g, ax, ai, an, int(normalise), ah, ay if y is not None else None,
                                                 C.c_void_p(loss.data_ptr()) if loss is not None else None, int(accumulate), st))
        return loss

    def evaluate(self, x, y, batch_size: Optional[int] = None, normalise: bool = False):
        """model.evaluate of every member on the same held-out split, one launch per batch for the whole group:
        [{'loss', 'mse', 'mae'}] per member, `loss` = `mse` as MLPEmulator.evaluate reports it (the first sum of the heads'
        epilogue: squared errors - SmoothL1 terms for a member built with loss='huber').  numpy / CPU inputs are moved to the
        group's device once."""
        import torch
        x = self.models[0]._to_device(x, self.models[0].input_length)
        y = self.models[0]._to_device(y, self.models[0].output_length)
        bs = min(batch_size or min(m.max_batch for m in self.models), min(m.max_batch for m in self.models))
        tot = torch.zeros((self.k, 2), dtype=torch.float32, device=self.device)
        for i, lo in enumerate(range(0, x.shape[0], bs)):
            hi = min(lo + bs, x.shape[0])
            self.forward_batch(x[lo:hi], y=y[lo:hi], normalise=normalise, loss=tot, accumulate=i > 0)
        s = tot.cpu().numpy().astype("float64")
        out = []
        for i, m in enumerate(self.models):
            d = s[i] / (m.output_length * x.shape[0])
            out.append({"loss": float(d[0]), "mse": float(d[0]), "mae": float(d[1])})
        return out

    def profile_step(self, x, y, lrs, row_idx=None, n=None, normalise: bool = False, active=None):
        """One grouped step with an event pair around each of its launches: {kind: (milliseconds, launches)}."""
        import torch
        ax, ay, ai, an, al, act = self._args(x, y, lrs, row_idx, n, active)
        kt = _lib.CsKernelTimes()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.cs_mlp_group_profile_step(self._g, ax, ay, ai, an, int(normalise), al, C.c_void_p(self._loss.data_ptr()), st,
                                                      C.byref(kt)))
        for m, a in zip(self.models, act):
            if a:
                m.iterations += 1
        return {k: (float(kt.ms[i]), int(kt.launches[i])) for i, k in enumerate(_lib.KERNEL_KINDS)}


def group_by_family(models: Sequence, max_members: int = 32) -> List[List[int]]:
    """Indices of `models` bucketed into groupable sets (one kernel family each, at most `max_members`); models on the
    per-layer path (family 0) come back as singletons."""
    buckets = {}
    out: List[List[int]] = []
    for i, m in enumerate(models):
        f = kernel_family(m)
        if f & 15 == 0:
            out.append([i])
            continue
        b = buckets.setdefault(f, [])
        b.append(i)
        if len(b) == max_members:
            out.append(b)
            buckets[f] = []
    out += [b for b in buckets.values() if b]
    return out
