"""Host side of the online-testing MLP on the MI355X engine.

Counterpart of `online_testing/baseline_models/MLP_v2rh/training/mlp.py` (class `MLP`: `Linear -> ReLU` per hidden
layer, `final_linear`, output pruning, ReLU on the last 8 outputs) and of the training step of
`train_mlp_h5loader.py` (:210-236 optimiser / schedulers / losses, :331-337 `training_step`), with the reference's
constructor arguments and torch `state_dict` layout, so checkpoints move both ways.  The arithmetic runs in
libclimsim_hip.so through `MLPEmulator` (direct-head topology, mse / mae / huber, torch-flavoured Adam); there is no
CPU execution path.  `export_wrapper` writes the TorchScript module E3SM loads for online coupling
(`online_testing/model_postprocessing/v2_nn_wrapper.ipynb` cell 5): that artefact is a CPU torch module by design.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

from .mlp import MLPEmulator


def output_keep_mask(out_dims: int, output_prune: bool, strato_lev_out: int) -> np.ndarray:
    """1 for the columns `MLP.forward` produces, 0 for `x[:, 60:60+lev] = 0` and the 120 / 180 / 240 blocks (mlp.py:56-61)."""
    keep = np.ones(out_dims, np.float32)
    if output_prune:
        for start in (60, 120, 180, 240):
            keep[start:start + strato_lev_out] = 0
    return keep


class StepLR:
    """torch.optim.lr_scheduler.StepLR, stepped once per epoch (train_mlp_h5loader.py:216-217)."""
    def __init__(self, lr, step_size, gamma):
        self.base, self.step_size, self.gamma = lr, step_size, gamma

    def __call__(self, epoch, val_loss=None):
        return self.base * self.gamma ** (epoch // self.step_size)


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR, closed form (train_mlp_h5loader.py:220-221)."""
    def __init__(self, lr, T_max, eta_min=0.0):
        self.base, self.T_max, self.eta_min = lr, T_max, eta_min

    def __call__(self, epoch, val_loss=None):
        return self.eta_min + (self.base - self.eta_min) * (1 + math.cos(math.pi * epoch / self.T_max)) / 2


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold 1e-4 rel) (train_mlp_h5loader.py:218-219)."""
    def __init__(self, lr, factor=0.1, patience=2):
        self.lr, self.factor, self.patience = lr, factor, patience
        self.best, self.bad = math.inf, 0

    def __call__(self, epoch, val_loss=None):
        if val_loss is not None:
            if val_loss < self.best * (1 - 1e-4):
                self.best, self.bad = val_loss, 0
            else:
                self.bad += 1
                if self.bad > self.patience:
                    self.lr, self.bad = self.lr * self.factor, 0
        return self.lr


class MLP:
    """`MLP(in_dims, out_dims, hidden_dims, layers, dropout=0., output_prune=False, strato_lev_out=15)` - mlp.py:28."""

    def __init__(self, in_dims: int, out_dims: int, hidden_dims, layers: int, dropout: float = 0.0,
                 output_prune: bool = False, strato_lev_out: int = 15, *, loss: str = "mse", n_relu: int = 8,
                 max_batch: int = 8192, device: Optional[int] = None, seed: Optional[int] = 0, eps: float = 1e-8, flags: int = 0,
                 dropout_seed: int = 0):
        if isinstance(hidden_dims, (list, tuple)):
            assert len(hidden_dims) == layers, "Length of hidden_dims should be equal to layers"      # mlp.py:33
            hidden = [int(h) for h in hidden_dims]
        else:
            hidden = [int(hidden_dims)] * layers
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        if output_prune and out_dims < 240 + strato_lev_out:
            raise ValueError("output pruning addresses columns up to 240 + strato_lev_out")
        self.in_dims, self.out_dims, self.hidden_dims, self.layers = in_dims, out_dims, hidden, layers
        self.output_prune, self.strato_lev_out, self.n_relu = output_prune, strato_lev_out, n_relu
        self.keep = output_keep_mask(out_dims, output_prune, strato_lev_out)
        self.engine = MLPEmulator(units=hidden, activation="relu", optimizer="AdamTorch", input_length=in_dims,
                                  output_length_lin=out_dims - n_relu, output_length_relu=n_relu, max_batch=max_batch,
                                  device=device, seed=None, epsilon=eps, flags=flags, direct_head=True, loss=loss,
                                  output_keep=self.keep if output_prune else None)
        self.loss_name = loss
        self.dropout = float(dropout)
        if self.dropout > 0.0:
            # training-mode nn.Dropout (mlp.py:39-44); torch's random stream is not reproducible by anyone else: the mask is a
            # counter hash of (dropout_seed, optimiser step, layer, row, column), shared with oracle/online_mlp_oracle.py
            self.engine.set_dropout(self.dropout, dropout_seed)
        if seed is not None:
            self.load_state_dict(self._torch_default_init(seed))

    # ---- parameters, torch layout: linears.{i}.0.weight (out,in) / .bias, final_linear.weight / .bias
    def _keys(self):
        return [f"linears.{i}.0" for i in range(self.layers)] + ["final_linear"]

    def _torch_default_init(self, seed) -> Dict[str, np.ndarray]:
        """nn.Linear.reset_parameters: weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
        rng = np.random.default_rng(seed)
        dims = [self.in_dims, *self.hidden_dims, self.out_dims]
        sd = {}
        for k, (fi, fo) in zip(self._keys(), zip(dims[:-1], dims[1:])):
            b = 1.0 / math.sqrt(fi)
            sd[k + ".weight"] = rng.uniform(-b, b, (fo, fi)).astype(np.float32)
            sd[k + ".bias"] = rng.uniform(-b, b, fo).astype(np.float32)
        return sd

    @staticmethod
    def _np(a):
        return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)

    def load_state_dict(self, sd):
        nl = self.out_dims - self.n_relu
        ws = []
        for k in self._keys():
            w, b = self._np(sd[k + ".weight"]).astype(np.float32).T, self._np(sd[k + ".bias"]).astype(np.float32)
            if k == "final_linear":
                ws += [w[:, :nl], b[:nl], w[:, nl:], b[nl:]]
            else:
                ws += [w, b]
        self.engine.set_weights([np.ascontiguousarray(a) for a in ws])

    def _to_state_dict(self, ws) -> Dict[str, np.ndarray]:
        sd = {}
        for i, k in enumerate(self._keys()[:-1]):
            sd[k + ".weight"], sd[k + ".bias"] = np.ascontiguousarray(ws[2 * i].T), ws[2 * i + 1]
        wl, bl, wr, br = ws[-4:]
        sd["final_linear.weight"] = np.ascontiguousarray(np.concatenate([wl, wr], axis=1).T)
        sd["final_linear.bias"] = np.concatenate([bl, br])
        return sd

    def state_dict(self) -> Dict[str, np.ndarray]:
        return self._to_state_dict(self.engine.get_weights())

    def gradients(self) -> Dict[str, np.ndarray]:
        """d(mean loss)/d(parameter) of the last `loss_grads` / `train_step`, state_dict layout (testing / inspection)."""
        return self._to_state_dict(self.engine.get_gradients(1.0 / (self.out_dims * self._last_n)))

    # ---- compute
    def forward(self, x, as_numpy: bool = False):
        """`model(x)`: (N, in_dims) -> (N, out_dims), pruned columns 0, last 8 >= 0."""
        return self.engine.predict(x, as_numpy=as_numpy)

    __call__ = forward

    def _loss_from_sums(self, sums, n) -> float:
        s = sums.detach().cpu().numpy() if hasattr(sums, "detach") else sums
        return float(s[1 if self.loss_name == "mae" else 0]) / (self.out_dims * n)

    def loss_grads(self, x, y):
        """forward + criterion(pred, target) + backward (no optimiser step); returns the loss."""
        x, y = self.engine._to_device(x, self.in_dims), self.engine._to_device(y, self.out_dims)
        self._last_n = x.shape[0]
        return self._loss_from_sums(self.engine.loss_grads(x, y), self._last_n)

    def train_step(self, x, y, lr: float):
        """`training_step` (train_mlp_h5loader.py:331-337) + optimizer.step(): returns the batch loss (synchronises)."""
        x, y = self.engine._to_device(x, self.in_dims), self.engine._to_device(y, self.out_dims)
        self._last_n = x.shape[0]
        return self._loss_from_sums(self.engine.train_on_batch(x, y, lr), self._last_n)

    def evaluate(self, x, y, batch_size: Optional[int] = None) -> float:
        """criterion(model(x), y) over a whole split (the validation loop, train_mlp_h5loader.py:427-457)."""
        import torch
        x, y = self.engine._to_device(x, self.in_dims), self.engine._to_device(y, self.out_dims)
        bs = min(batch_size or self.engine.max_batch, self.engine.max_batch)
        acc = torch.zeros(2, dtype=torch.float32, device=self.engine.device)
        for lo in range(0, x.shape[0], bs):
            self.engine.forward_batch(x[lo:lo + bs], y=y[lo:lo + bs], loss=acc, accumulate=True)
        return self._loss_from_sums(acc, x.shape[0])

    def fit(self, x, y, batch_size: int = 1024, epochs: int = 1, learning_rate: float = 1e-4, scheduler=None,
            validation_data=None, shuffle_seed: int = 0, verbose: bool = False):
        """The epoch loop of train_mlp_h5loader.py:348-470: shuffled batches (DataLoader(shuffle=True), last partial
        batch kept), one optimiser step each, validation + scheduler.step() per epoch.  `scheduler`: None, or one of
        StepLR / CosineAnnealingLR / ReduceLROnPlateau above (called with the epoch index and the validation loss)."""
        import torch
        x, y = self.engine._to_device(x, self.in_dims), self.engine._to_device(y, self.out_dims)
        gen = torch.Generator(device=self.engine.device)
        gen.manual_seed(shuffle_seed)
        hist = {"loss": [], "val_loss": [], "lr": []}
        lr = learning_rate
        for ep in range(epochs):
            perm = torch.randperm(x.shape[0], device=self.engine.device, generator=gen)
            nsteps = (x.shape[0] + batch_size - 1) // batch_size
            sums = torch.zeros((nsteps, 2), dtype=torch.float32, device=self.engine.device)   # one slot per step, written by the engine
            for k, lo in enumerate(range(0, x.shape[0], batch_size)):
                self.engine.train_on_batch(x, y, lr, row_idx=perm[lo:lo + batch_size], loss=sums[k])
            hist["loss"].append(self._loss_from_sums(sums.sum(dim=0), x.shape[0]))
            hist["lr"].append(lr)
            vl = self.evaluate(*validation_data) if validation_data is not None else None
            hist["val_loss"].append(vl)
            if scheduler is not None:
                lr = scheduler(ep + 1, vl)
            if verbose:
                print(f"epoch {ep + 1}: loss {hist['loss'][-1]:.6f} val {vl} lr {hist['lr'][-1]:.3e}")
        return hist

    # ---- export for online coupling
    def export_wrapper(self, path: str, input_sub: Sequence[float], input_div: Sequence[float], out_scale: Sequence[float],
                       lbd_qc: Optional[Sequence[float]] = None, lbd_qi: Optional[Sequence[float]] = None,
                       qn_prune_levels: int = 15, rh_clip=(0.0, 1.2), post_prune=((60, 75), (120, 148), (180, 195), (240, 255), (300, 315))):
        """TorchScript `normalise -> model -> zero pruned outputs -> / out_scale` module for E3SM's torch coupler:
        `NewModel` + `save_wrapper` of online_testing/model_postprocessing/v2_nn_wrapper.ipynb (cells 5-6).  The qc / qi
        exponential transforms apply when `lbd_qc` / `lbd_qi` are given (v4 / v5 variable sets, input columns 120:180 and
        180:240), the humidity clip when `rh_clip` is not None.  Returns the scripted module (also saved to `path`)."""
        import torch
        from torch import nn

        sd = self.state_dict()
        keep = torch.from_numpy(self.keep.copy())
        n_relu = self.n_relu

        class Core(nn.Module):                                        # mlp.py:28-67 with the trained weights
            def __init__(self, keys):
                super().__init__()
                self.linears = nn.ModuleList()
                for k in keys[:-1]:
                    w = sd[k + ".weight"]
                    lin = nn.Linear(w.shape[1], w.shape[0])
                    lin.weight.data, lin.bias.data = torch.from_numpy(w.copy()), torch.from_numpy(sd[k + ".bias"].copy())
                    self.linears.append(lin)
                w = sd["final_linear.weight"]
                self.final_linear = nn.Linear(w.shape[1], w.shape[0])
                self.final_linear.weight.data = torch.from_numpy(w.copy())
                self.final_linear.bias.data = torch.from_numpy(sd["final_linear.bias"].copy())
                self.register_buffer("keep", keep)
                self.n_relu = n_relu

            def forward(self, x):
                for lin in self.linears:
                    x = torch.relu(lin(x))
                x = self.final_linear(x) * self.keep
                return torch.cat([x[:, :-self.n_relu], torch.relu(x[:, -self.n_relu:])], dim=1)

        class Wrapper(nn.Module):                                     # v2_nn_wrapper.ipynb cell 5: NewModel
            def __init__(self, core):
                super().__init__()
                self.core = core
                f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
                self.register_buffer("input_sub", f(input_sub))
                self.register_buffer("input_div", f(input_div))
                self.register_buffer("out_scale", f(out_scale))
                self.has_q = lbd_qc is not None and lbd_qi is not None
                self.register_buffer("lbd_qc", f(lbd_qc if lbd_qc is not None else [1.0]))
                self.register_buffer("lbd_qi", f(lbd_qi if lbd_qi is not None else [1.0]))
                self.qn = qn_prune_levels
                self.clip = rh_clip is not None
                self.lo, self.hi = (float(rh_clip[0]), float(rh_clip[1])) if rh_clip is not None else (0.0, 0.0)
                post = torch.ones(len(out_scale))
                for a, b in post_prune:
                    post[a:b] = 0
                self.register_buffer("post", post)

            def forward(self, x):
                x = x.clone()
                if self.has_q:
                    x[:, 120:180] = 1 - torch.exp(-x[:, 120:180] * self.lbd_qc)
                    x[:, 180:240] = 1 - torch.exp(-x[:, 180:240] * self.lbd_qi)
                x = (x - self.input_sub) / self.input_div
                x = torch.where(torch.isnan(x) | torch.isinf(x), torch.zeros_like(x), x)
                if self.has_q:
                    x[:, 120:120 + self.qn] = 0
                    x[:, 180:180 + self.qn] = 0
                if self.clip:
                    x[:, 60:120] = torch.clamp(x[:, 60:120], self.lo, self.hi)
                y = self.core(x)
                return y * self.post / self.out_scale

        wrapped = Wrapper(Core(self._keys())).eval()
        for prm in wrapped.parameters():
            prm.requires_grad_(False)                                 # an inference artefact
        mod = torch.jit.script(wrapped)
        mod.save(path)
        return mod
