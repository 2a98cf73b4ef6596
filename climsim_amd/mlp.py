"""Host side of the MI355X MLP engine: the counterpart of the reference's Keras model object.

Mirrors what `baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py` does
with Keras - `build_model` (:79-167), `model.fit(..., callbacks=[best, last, CSVLogger,
EarlyStopping])` (:253-285) - and `model.predict` of step3_inference.ipynb, with the same names
and argument meaning where they exist (fit / predict / evaluate / get_weights / set_weights /
count_params, `units`, `activation`, `optimizer`, `batch_size`, `epochs`, `validation_data`).

All arithmetic happens in libclimsim_hip.so (include/climsim_hip.h); torch is used for device
memory, streams and torch.distributed (RCCL) only.  There is no CPU execution path: without the
library or without a GPU every compute call raises.
"""
from __future__ import annotations

import csv
import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


# ------------------------------------------------------------------------------- LR schedules
@dataclass
class CyclicalLearningRate:
    """tfa.optimizers.CyclicalLearningRate with scale_fn=1/2**(cycle-1), scale_mode='cycle'
    (triangular2) - step2_retrain.py:140-148.  `step_size` = 2*steps_per_epoch there."""
    initial_learning_rate: float = 2.5e-4
    maximal_learning_rate: float = 2.5e-3
    step_size: float = 16

    def __call__(self, step: int) -> float:
        cycle = math.floor(1 + step / (2 * self.step_size))
        x = abs(step / self.step_size - 2 * cycle + 1)
        return (self.initial_learning_rate
                + (self.maximal_learning_rate - self.initial_learning_rate) * max(0.0, 1 - x) / 2.0 ** (cycle - 1))


@dataclass
class ConstantLearningRate:
    learning_rate: float = 1e-3

    def __call__(self, step: int) -> float:
        return self.learning_rate


def glorot_uniform_weights(n_in, units, n_out_lin, n_out_relu, seed, direct_head=False):
    """Keras-ordered initial weights: glorot_uniform kernels, zero biases (Dense defaults)."""
    rng = np.random.default_rng(seed)
    dims = [n_in, *units] + ([] if direct_head else [n_out_lin + n_out_relu])
    shapes = [(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    shapes += [(dims[-1], n_out_lin), (dims[-1], n_out_relu)]
    ws = []
    for k, n in shapes:
        lim = math.sqrt(6.0 / (k + n))
        ws.append(rng.uniform(-lim, lim, size=(k, n)).astype(np.float32))
        ws.append(np.zeros(n, dtype=np.float32))
    return ws


def _torch():
    import torch
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------- the model
class MLPEmulator:
    """ClimSim baseline MLP (input_length -> units... -> output_length -> [lin || relu]) on one MI355X:
    124 -> ... -> 128 -> [120 || 8] for the v1 variable set (layer-chain kernels), e.g. 425 -> ... -> 368 -> [360 || 8]
    for v2 (hpo_baseline_v2.py:58-101; wide layer-chain kernels)."""

    def __init__(self, units: Sequence[int] = (512, 512, 512, 512, 512), activation: str = "leakyrelu",
                 optimizer: str = "Adam", input_length: int = 124, output_length_lin: int = 120,
                 output_length_relu: int = 8, alpha: float = 0.15, max_batch: int = 8192,
                 device: Optional[int] = None, seed: Optional[int] = 0, beta_1: float = 0.9,
                 beta_2: float = 0.999, epsilon: float = 1e-7, rho: float = 0.9, flags: int = 0,
                 direct_head: bool = False, loss: str = "mse", output_keep=None, cooperative: bool = False):
        """`direct_head`, `loss` ('mse' | 'mae' | 'huber'), `output_keep` (1/0 per output column) and optimizer
        'AdamTorch' are the pieces of the online-testing MLP (climsim_amd/online_mlp.py); the baseline models leave
        them at their defaults.  `cooperative`: training steps of up to 2048 columns split every 32-row tile over 8 / 4
        workgroups (CS_FLAG_COOP, csrc/coop.h: 1.3x at batch 1024) - only when this process is the one user of the GPU
        while such launches run (their workgroups wait for one another; a side-stream kernel or a second process can starve
        them).  Opt-in for that reason: a library cannot know that it owns the device.  A wait that runs out is counted by
        the kernel in host-mapped memory: every later call on the model raises EngineError, `check()` forces the test
        behind a synchronisation (fit() calls it at the end of every epoch), `coop_timeouts` reads the counter."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.EngineError("MLPEmulator needs a ROCm GPU (no CPU fallback)")
        if activation not in _lib.ACT:
            raise ValueError(f"activation must be one of {list(_lib.ACT)}")
        if optimizer not in _lib.OPT:
            raise ValueError(f"optimizer must be one of {list(_lib.OPT)}")
        if loss not in _lib.LOSS:
            raise ValueError(f"loss must be one of {list(_lib.LOSS)}")
        self.lib = _lib.load()
        self.direct_head, self.loss_name = bool(direct_head), loss
        if self.direct_head:
            flags |= _lib.FLAG_DIRECT_HEAD
        if cooperative:
            flags |= _lib.CS_FLAG_COOP
        self.cooperative = bool(cooperative)
        self.units = tuple(int(u) for u in units)
        self.activation, self.optimizer_name = activation, optimizer
        self.input_length, self.output_length = input_length, output_length_lin + output_length_relu
        self.output_length_lin, self.output_length_relu = output_length_lin, output_length_relu
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.max_batch = int(max_batch)
        cfg = _lib.CsMlpCfg()
        cfg.n_in, cfg.n_hidden = input_length, len(self.units)
        for i, u in enumerate(self.units):
            cfg.hidden[i] = u
        cfg.n_out_lin, cfg.n_out_relu = output_length_lin, output_length_relu
        cfg.act, cfg.alpha, cfg.optimizer = _lib.ACT[activation], alpha, _lib.OPT[optimizer]
        cfg.beta1, cfg.beta2, cfg.eps, cfg.rho = beta_1, beta_2, epsilon, rho
        cfg.max_batch, cfg.device, cfg.flags = self.max_batch, self.device_index, flags
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cs_mlp_create(C.byref(self._h), C.byref(cfg)))
        self._n_params = int(self.lib.cs_mlp_num_params(self._h))
        self.set_head_options(loss, output_keep)
        self._loss = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._grad_tensor = None
        self.gradient_tensor()          # gradients always live in a torch tensor (all-reduce payload)
        self.iterations = 0
        self.stop_training = False
        if seed is not None:
            self.set_weights(glorot_uniform_weights(input_length, self.units, output_length_lin,
                                                    output_length_relu, seed, self.direct_head))

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.cs_mlp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def check(self):
        """Synchronise the current stream and raise EngineError if a cooperative launch (CS_FLAG_COOP) has timed out
        since the model was built - everything computed since then, optimiser state included, is invalid (cs_mlp_check)."""
        _lib.check(self.lib.cs_mlp_check(self._h, self._stream()))

    @property
    def coop_timeouts(self) -> int:
        """Bounded waits of cooperative launches that ran out so far (0 for a healthy model; no synchronisation)."""
        return int(self.lib.cs_mlp_coop_timeouts(self._h))

    # ---- weights
    def count_params(self) -> int:
        return self._n_params

    def set_head_options(self, loss: str = "mse", output_keep=None):
        """Training loss and output pruning (cs_mlp_set_head_options): `output_keep` holds 1 for the output columns
        the model produces and 0 for the ones forced to zero (MLP_v2rh/training/mlp.py:56-61)."""
        if loss not in _lib.LOSS:
            raise ValueError(f"loss must be one of {list(_lib.LOSS)}")
        keep = None
        if output_keep is not None:
            keep = np.ascontiguousarray(output_keep, dtype=np.float32)
            if keep.shape != (self.output_length,):
                raise ValueError(f"output_keep must have {self.output_length} entries")
        _lib.check(self.lib.cs_mlp_set_head_options(self._h, _lib.LOSS[loss], keep.ctypes.data_as(C.c_void_p) if keep is not None else None,
                                                    0 if keep is None else keep.size))
        self.loss_name, self.output_keep = loss, keep

    def set_dropout(self, rate: float, seed: int = 0):
        """nn.Dropout(rate) behind every hidden layer while training (cs_mlp_set_dropout; ReLU stacks, wide chain);
        prediction / evaluation stay in eval mode."""
        _lib.check(self.lib.cs_mlp_set_dropout(self._h, float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF))
        self.dropout, self.dropout_seed = float(rate), int(seed)

    def _shapes(self):
        dims = [self.input_length, *self.units] + ([] if self.direct_head else [self.output_length])
        sh = []
        for i in range(len(dims) - 1):
            sh += [(dims[i], dims[i + 1]), (dims[i + 1],)]
        sh += [(dims[-1], self.output_length_lin), (self.output_length_lin,),
               (dims[-1], self.output_length_relu), (self.output_length_relu,)]
        return sh

    def _split(self, flat):
        out, at = [], 0
        for s in self._shapes():
            n = int(np.prod(s))
            out.append(flat[at:at + n].reshape(s).copy())
            at += n
        return out

    def _flatten(self, weights):
        shapes = self._shapes()
        if len(weights) != len(shapes):
            raise ValueError(f"expected {len(shapes)} arrays (Keras order), got {len(weights)}")
        for w, s in zip(weights, shapes):
            if tuple(np.shape(w)) != tuple(s):
                raise ValueError(f"weight shape {np.shape(w)} does not match {s}")
        return np.ascontiguousarray(np.concatenate([np.asarray(w, dtype=np.float32).ravel() for w in weights]))

    def set_weights(self, weights: List[np.ndarray]):
        """model.set_weights: Keras order [W0,b0,...,W_lin,b_lin,W_relu,b_relu], kernels (in,out)."""
        flat = self._flatten(weights)
        _lib.check(self.lib.cs_mlp_set_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size, self._stream()))

    def get_weights(self) -> List[np.ndarray]:
        flat = np.empty(self._n_params, dtype=np.float32)
        _lib.check(self.lib.cs_mlp_get_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size, self._stream()))
        return self._split(flat)

    def get_optimizer_state(self):
        m = np.empty(self._n_params, dtype=np.float32)
        v = np.empty(self._n_params, dtype=np.float32)
        it = C.c_int64()
        _lib.check(self.lib.cs_mlp_get_opt_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                                 m.size, C.byref(it), self._stream()))
        return self._split(m), self._split(v), int(it.value)

    def set_optimizer_state(self, m, v, iterations):
        fm, fv = self._flatten(m), self._flatten(v)
        _lib.check(self.lib.cs_mlp_set_opt_state(self._h, fm.ctypes.data_as(C.c_void_p), fv.ctypes.data_as(C.c_void_p),
                                                 fm.size, int(iterations), self._stream()))
        self.iterations = int(iterations)

    def save_weights(self, path: str):
        """Checkpoint = weights + optimiser state.  `*.h5`: Keras' legacy-H5 layout, the format of the reference's
        ModelCheckpoint(save_weights_only=False) files (step2_retrain.py:253-261; climsim_amd/keras_h5.py) - a
        reference-built Keras model can `load_weights` it; anything else: .npz (Keras-ordered arrays w0..., m0..., v0...)."""
        ws = self.get_weights()
        m, v, it = self.get_optimizer_state()
        if path.endswith((".h5", ".hdf5")):
            if self.direct_head:
                raise ValueError("the Keras .h5 layout describes the baseline MLP (Dense(output_length) + two heads); use .npz")
            from .keras_h5 import save_keras_h5
            save_keras_h5(path, ws, self.activation, full_model=True, optimizer_state={"m": m, "v": v, "iterations": it})
            return
        blob = {f"w{i}": a for i, a in enumerate(ws)}
        blob.update({f"m{i}": a for i, a in enumerate(m)})
        blob.update({f"v{i}": a for i, a in enumerate(v)})
        blob["iterations"] = np.int64(it)
        blob["units"] = np.asarray(self.units)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **blob)
        os.replace(tmp, path)

    def load_weights(self, path: str, with_optimizer: bool = True):
        """Load a checkpoint written by `save_weights`, or a Keras `.h5` file of the reference (model.save / save_weights /
        ModelCheckpoint: e.g. the published baseline_models/MLP/model/*.best.h5) - weights are taken in `layer_names` order."""
        if path.endswith((".h5", ".hdf5")):
            from .keras_h5 import load_keras_h5
            ws, opt = load_keras_h5(path, with_optimizer=True)
            self.set_weights(ws)
            if with_optimizer and opt is not None:
                self.set_optimizer_state(opt["m"], opt["v"], opt["iterations"])
            return
        z = np.load(path)
        n = len(self._shapes())
        self.set_weights([z[f"w{i}"] for i in range(n)])
        if with_optimizer and "m0" in z.files:
            self.set_optimizer_state([z[f"m{i}"] for i in range(n)], [z[f"v{i}"] for i in range(n)],
                                     int(z["iterations"]))

    def set_norm(self, input_sub, input_div):
        """data_utils.save_norm() vectors for in-kernel (x-sub)/div normalisation of raw inputs."""
        s = np.ascontiguousarray(input_sub, dtype=np.float32)
        d = np.ascontiguousarray(input_div, dtype=np.float32)
        if s.shape != (self.input_length,) or d.shape != (self.input_length,):
            raise ValueError("norm vectors must have input_length entries")
        _lib.check(self.lib.cs_mlp_set_norm(self._h, s.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p)))

    # ---- device helpers
    def _to_device(self, a, cols):
        torch = _torch()
        if a is None:
            return None
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if a.ndim != 2 or a.shape[1] != cols:
            raise ValueError(f"expected a (N,{cols}) array, got {tuple(a.shape)}")
        return a

    # ---- compute entry points
    def forward_batch(self, x, yhat=None, y=None, row_idx=None, n=None, normalise=False, loss=None, accumulate=False):
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        loss = self._loss if (loss is None and y is not None) else loss
        _lib.check(self.lib.cs_mlp_forward(self._h, _ptr(x), _ptr(row_idx), n, int(normalise), _ptr(yhat), _ptr(y),
                                           _ptr(loss), int(accumulate), self._stream()))
        return loss

    def loss_grads(self, x, y, row_idx=None, n=None, normalise=False, loss=None, accumulate=False):
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        loss = self._loss if loss is None else loss
        _lib.check(self.lib.cs_mlp_loss_grads(self._h, _ptr(x), _ptr(y), _ptr(row_idx), n, int(normalise), _ptr(loss),
                                              int(accumulate), self._stream()))
        return loss

    def apply_gradients(self, lr: float, grad_scale: float):
        _lib.check(self.lib.cs_mlp_apply(self._h, float(lr), float(grad_scale), self._stream()))
        self.iterations += 1

    def train_on_batch(self, x, y, lr: float, row_idx=None, n=None, normalise=False, loss=None):
        """One optimiser step on one batch (Model.train_step).  Returns the device tensor
        [sum sq err, sum abs err]; divide by 128*n for mse / mae.  Asynchronous."""
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        loss = self._loss if loss is None else loss
        _lib.check(self.lib.cs_mlp_train_step(self._h, _ptr(x), _ptr(y), _ptr(row_idx), n, int(normalise), float(lr),
                                              _ptr(loss), self._stream()))
        self.iterations += 1
        return loss

    def profile_step(self, x, y, lr: float, row_idx=None, n=None, normalise=False):
        """One train step with a HIP-event pair around every kernel launch (on the launch stream).
        Returns {kind: (milliseconds, launches)}."""
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        kt = _lib.CsKernelTimes()
        _lib.check(self.lib.cs_mlp_profile_step(self._h, _ptr(x), _ptr(y), _ptr(row_idx), n, int(normalise), float(lr),
                                                _ptr(self._loss), self._stream(), C.byref(kt)))
        self.iterations += 1
        return {k: (float(kt.ms[i]), int(kt.launches[i])) for i, k in enumerate(_lib.KERNEL_KINDS)}

    def gradient_tensor(self):
        """Flat float32 gradient buffer as a torch tensor (allocated by torch and bound into the
        engine) - the payload of the one-per-step RCCL all-reduce.  Internal parameter order (heads fused,
        output layers padded to a multiple of 128 columns), so it may be longer than count_params()."""
        if self._grad_tensor is None:
            torch = _torch()
            ptr, n = C.c_void_p(), C.c_int64(0)
            _lib.check(self.lib.cs_mlp_grad_buffer(self._h, C.byref(ptr), C.byref(n)))
            self._grad_tensor = torch.zeros(int(n.value), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.cs_mlp_set_grad_buffer(self._h, _ptr(self._grad_tensor)))
        return self._grad_tensor

    def bind_gradient_tensor(self, tensor):
        """Make `tensor` (float32, device, at least gradient_tensor().numel() elements, 16-byte aligned) the engine's flat
        gradient buffer - e.g. the exchange buffer of the one-shot all-reduce (climsim_amd/dp.py: IpcComm)."""
        n = self.gradient_tensor().numel()
        if tensor.numel() < n or tensor.dtype != _torch().float32 or not tensor.is_cuda:
            raise ValueError("gradient buffer must be a float32 device tensor of at least %d elements" % n)
        tensor.zero_()
        _lib.check(self.lib.cs_mlp_set_grad_buffer(self._h, _ptr(tensor)))
        self._grad_tensor = tensor[:n]

    def get_gradients(self, grad_scale: float = 1.0) -> List[np.ndarray]:
        """Gradients of the last loss_grads call, Keras order (testing / inspection)."""
        self.gradient_tensor()
        flat = np.empty(self._n_params, np.float32)
        _lib.check(self.lib.cs_mlp_get_grads(self._h, flat.ctypes.data_as(C.c_void_p), flat.size, self._stream()))
        return self._split(flat * np.float32(grad_scale))

    # ---- Keras-like API
    def _forward_chunk(self, batch_size: Optional[int]) -> int:
        """Rows per cs_mlp_forward call of predict / evaluate: the caller's `batch_size`, else 65536 where the engine takes calls
        beyond max_batch (layer-chain paths: nothing is kept per row; tall tiles fill the chip from 32768 rows), else max_batch."""
        limit = int(self.lib.cs_mlp_forward_limit(self._h))
        return max(1, min(int(batch_size) if batch_size else max(self.max_batch, min(65536, limit)), limit))

    def predict(self, x, batch_size: Optional[int] = None, normalise: bool = False, as_numpy: bool = True):
        """model.predict: (N,124) -> (N,128) float32 in scaled output space."""
        torch = _torch()
        x = self._to_device(x, self.input_length)
        bs = self._forward_chunk(batch_size)
        out = torch.empty((x.shape[0], self.output_length), dtype=torch.float32, device=self.device)
        for lo in range(0, x.shape[0], bs):
            hi = min(lo + bs, x.shape[0])
            self.forward_batch(x[lo:hi], yhat=out[lo:hi], normalise=normalise)
        return out.cpu().numpy() if as_numpy else out

    def evaluate(self, x, y, batch_size: Optional[int] = None, normalise: bool = False, accuracy: bool = False):
        """model.evaluate: {'loss','mse','mae'} over the whole set (loss = mse).  `accuracy` adds Keras' `accuracy`
        metric of compile(metrics=['mse','mae','accuracy']) (step2_retrain.py:160-162), which for a 128-column target is
        categorical accuracy: the share of rows with argmax(y_true) == argmax(y_pred) (cs_categorical_accuracy)."""
        torch = _torch()
        x, y = self._to_device(x, self.input_length), self._to_device(y, self.output_length)
        bs = self._forward_chunk(batch_size)
        tot = torch.zeros(2, dtype=torch.float32, device=self.device)
        hits = torch.zeros(1, dtype=torch.int64, device=self.device) if accuracy else None
        yhat = torch.empty((bs, self.output_length), dtype=torch.float32, device=self.device) if accuracy else None
        for i, lo in enumerate(range(0, x.shape[0], bs)):
            hi = min(lo + bs, x.shape[0])
            self.forward_batch(x[lo:hi], yhat=yhat, y=y[lo:hi], normalise=normalise, loss=tot, accumulate=i > 0)
            if accuracy:
                _lib.check(self.lib.cs_categorical_accuracy(_ptr(yhat), _ptr(y[lo:hi]), hi - lo, self.output_length,
                                                            _ptr(hits), 1, self._stream()))
        s = tot.cpu().numpy().astype(np.float64) / (self.output_length * x.shape[0])
        if self.cooperative:
            self.check()                 # the copy above synchronised: a time-out of an earlier training step surfaces here
        out = {"loss": float(s[0]), "mse": float(s[0]), "mae": float(s[1])}
        if accuracy:
            out["accuracy"] = float(hits.item()) / x.shape[0]
        return out

    def fit(self, x, y, batch_size: int = 1024, epochs: int = 1, validation_data=None, learning_rate=None,
            shuffle: bool = True, seed: int = 0, normalise: bool = False, csv_log: Optional[str] = None,
            checkpoint_best: Optional[str] = None, checkpoint_last: Optional[str] = None,
            early_stopping_patience: Optional[int] = None, steps_per_epoch: Optional[int] = None,
            distributed: bool = False, verbose: int = 0, train_accuracy: Optional[bool] = None):
        """model.fit on an HBM-resident split.

        The reference pipeline (step2_retrain.py:266-277) streams files through a windowed
        shuffle(384*30); here the whole split lives in device memory, each epoch draws a global
        permutation on device and every step hands a slice of it to the engine, which gathers the
        rows inside its first kernel.  With `distributed` (torch.distributed initialised, one
        process per GPU) `batch_size` is the GLOBAL batch: every rank takes the slice
        rank::world of each global batch, gradients are summed with ONE all-reduce per step and
        scaled by 1/(128*global_batch) in the optimiser kernel.
        Returns the history dict (loss/mse/mae[, val_*], lr per epoch), like History.history.
        """
        torch = _torch()
        x, y = self._to_device(x, self.input_length), self._to_device(y, self.output_length)
        val = None
        if validation_data is not None:
            val = (self._to_device(validation_data[0], self.input_length),
                   self._to_device(validation_data[1], self.output_length))
        sched = ConstantLearningRate(1e-3) if learning_rate is None else learning_rate
        if not callable(sched):
            sched = ConstantLearningRate(float(sched))
        from .dp import DataParallel, shard_of_batch
        dist = None
        if distributed:
            import torch.distributed as dist
        dp = DataParallel(self, dist, self.output_length)
        rank, world = dp.rank, dp.world
        if distributed:
            dp.broadcast_weights()
        if batch_size % world:
            raise ValueError("global batch_size must be divisible by the world size")
        local_bs = batch_size // world
        if local_bs > self.max_batch:
            raise ValueError(f"per-GPU batch {local_bs} exceeds max_batch {self.max_batch}")
        n = x.shape[0]
        steps = steps_per_epoch or (n // batch_size)
        if steps < 1:
            raise ValueError("dataset smaller than one batch")
        gen = torch.Generator(device=self.device)
        history = {k: [] for k in ("loss", "mse", "mae", "lr")}
        if val is not None:
            history.update({k: [] for k in ("val_loss", "val_mse", "val_mae", "val_accuracy")})
        # keras.callbacks.CSVLogger writes `epoch` + the log keys in sorted order (step2_retrain.py:262; the reference's own
        # logs: baseline_models/ED/model/ED_ClimSIM_1_3.csv:1 `epoch,accuracy,loss,lr,mae,mse,val_accuracy,val_loss,val_mae,
        # val_mse`) and "NA" for a key without a value.  The training-pass `accuracy` (argmax match over the 128 outputs -
        # meaningless for a regression, but a column of the reference's log) is counted by the engine when asked for
        # (cs_mlp_set_train_accuracy: on by default when a CSV log is written), `val_accuracy` on the validation pass.
        want_acc = bool(csv_log) if train_accuracy is None else bool(train_accuracy)
        acc_count = torch.zeros(1, dtype=torch.int64, device=self.device) if want_acc else None
        if want_acc:
            history["accuracy"] = []
        csv_keys = sorted({"accuracy", *history.keys()})
        best, wait = math.inf, 0
        writer = f = None
        self.stop_training = False
        try:
            # everything that can raise sits inside the try: the engine holds the counter's DEVICE POINTER from here on, and the
            # `finally` below takes it back before `acc_count` can be freed (a log file that cannot be opened used to leave it
            # dangling: the next train_on_batch would have added to freed memory)
            if want_acc:
                _lib.check(self.lib.cs_mlp_set_train_accuracy(self._h, _ptr(acc_count)))
            if csv_log and rank == 0:
                new = not os.path.exists(csv_log)
                f = open(csv_log, "a", newline="")
                writer = csv.writer(f)
                if new:
                    writer.writerow(["epoch", *csv_keys])
            for epoch in range(epochs):
                gen.manual_seed(seed + epoch)            # identical permutation on every rank
                perm = torch.randperm(n, device=self.device, generator=gen) if shuffle else torch.arange(n, device=self.device)
                lr = sched(self.iterations)
                # one [sum sq err, sum abs err] slot per step, written by the engine: nothing is launched to add them up
                step_loss = torch.zeros((steps, 2), dtype=torch.float32, device=self.device)
                if want_acc:
                    acc_count.zero_()
                for s in range(steps):
                    lr = sched(self.iterations)
                    if distributed:
                        dp.train_step(x, y, perm, s, batch_size, lr, loss=step_loss[s], normalise=normalise)
                    else:
                        idx = shard_of_batch(perm, s, batch_size, 0, 1)
                        self.train_on_batch(x, y, lr, row_idx=idx, normalise=normalise, loss=step_loss[s])
                epoch_sum = step_loss.sum(dim=0)
                if distributed:
                    dist.all_reduce(epoch_sum)
                tr = epoch_sum.cpu().numpy().astype(np.float64) / (self.output_length * batch_size * steps)
                if self.cooperative:
                    self.check()         # a cooperative launch that timed out during the epoch fails the epoch, not the next checkpoint
                row = {"loss": float(tr[0]), "mse": float(tr[0]), "mae": float(tr[1]), "lr": float(lr)}
                if want_acc:
                    hits = acc_count.clone()
                    if distributed:
                        dist.all_reduce(hits)
                    row["accuracy"] = float(hits.item()) / (batch_size * steps)
                if val is not None:
                    ev = self.evaluate(val[0], val[1], normalise=normalise, accuracy=True)
                    row.update({"val_loss": ev["loss"], "val_mse": ev["mse"], "val_mae": ev["mae"], "val_accuracy": ev["accuracy"]})
                for k, v in row.items():
                    history[k].append(v)
                if writer:
                    writer.writerow([epoch, *[row.get(k, "NA") for k in csv_keys]])
                    f.flush()
                if verbose and rank == 0:
                    print(f"epoch {epoch + 1}/{epochs} " + " ".join(f"{k}={v:.6g}" for k, v in row.items()), flush=True)
                monitor = row.get("val_loss", row["loss"])
                if not math.isfinite(monitor):
                    raise FloatingPointError(f"non-finite loss at epoch {epoch}")
                if rank == 0 and checkpoint_last:
                    self.save_weights(checkpoint_last)
                if monitor < best:
                    best, wait = monitor, 0
                    if rank == 0 and checkpoint_best:
                        self.save_weights(checkpoint_best)
                else:
                    wait += 1
                    if early_stopping_patience is not None and wait >= early_stopping_patience:
                        self.stop_training = True
                        break
        finally:
            if f is not None:
                f.close()
            self.lib.cs_mlp_set_train_accuracy(self._h, None)      # unconditionally: no pointer of this call survives it
            dp.close()                     # the engine's RCCL communicator never outlives the call, also on an exception
        return history


def build_model(units, activation="leakyrelu", optimizer="RAdam", batch_size=3072, **kw):
    """Counterpart of step2_retrain.build_model for explicit hyper-parameters (the reference reads
    them from step1_results.csv): returns (model, batch_size)."""
    return MLPEmulator(units=units, activation=activation, optimizer=optimizer,
                       max_batch=kw.pop("max_batch", max(batch_size, 8192)), **kw), batch_size
