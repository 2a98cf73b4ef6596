"""Host side of the level-axis CNN engine: counterpart of the Keras model built by
`CNNHyperModel.build` and trained by `main()` (baseline_models/CNN/training/hpo_train.py:124-236,
:294-368).  Same conventions as `climsim_amd.mlp`: arithmetic in libclimsim_hip.so, torch only for
device memory, streams and torch.distributed, no CPU fallback."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import os

import numpy as np

from . import _lib
from .mlp import CyclicalLearningRate, ConstantLearningRate

OPTIMIZERS = {"Adam": 0, "SGD": 3}
LOSSES = {"mean_absolute_error": 0, "mae_adjusted": 0, "mae": 0, "mse": 1, "mse_adjusted": 1}


def cnn_learning_rate(depth: int = 12):
    """The schedule of hpo_train.py:203-213: tfa CyclicalLearningRate(1e-4, 1e-3, triangular2) whose half-cycle
    is 2 * (10091520 // hp_depth) steps (the reference divides the row count by the DEPTH, not the batch)."""
    return CyclicalLearningRate(1e-4, 1e-3, 2 * (10091520 // depth))


def _shapes(depth, channels, c_in=6, c_out=10, n_lin=2, k=3):
    sh = []
    for b in range(depth):
        cin = c_in if b == 0 else channels
        sh += [(k, cin, channels), (channels,), (k, channels, channels), (channels,), (1, cin, channels), (channels,)]
    sh += [(1, channels, c_out), (c_out,), (c_out, n_lin), (n_lin,), (c_out, c_out - n_lin), (c_out - n_lin,)]
    return sh


def glorot_uniform_cnn(depth=12, channels=406, seed=0):
    """Keras default initialisation (glorot_uniform kernels with fan = receptive field * channels, zero biases)."""
    rng = np.random.default_rng(seed)
    ws = []
    for s in _shapes(depth, channels):
        if len(s) == 1:
            ws.append(np.zeros(s, np.float32))
        else:
            rf = int(np.prod(s[:-2])) if len(s) == 3 else 1
            lim = math.sqrt(6.0 / (rf * s[-2] + rf * s[-1]))
            ws.append(rng.uniform(-lim, lim, s).astype(np.float32))
    return ws


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class CNNEmulator:
    """(B,60,6) -> (B,60,10) ResNet-style 1-D CNN over the level axis on one MI355X."""

    def __init__(self, depth: int = 12, channel_width: int = 406, kernel_width: int = 3, max_batch: int = 512,
                 device: Optional[int] = None, trainable: bool = False, optimizer: str = "Adam",
                 loss: str = "mean_absolute_error", dropout: float = 0.175, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, seed: int = 0, init_seed: Optional[int] = None, tile128: bool = False):
        import torch
        if not torch.cuda.is_available():
            raise _lib.EngineError("CNNEmulator needs a ROCm GPU (no CPU fallback)")
        if optimizer not in OPTIMIZERS:
            raise ValueError(f"optimizer must be one of {sorted(OPTIMIZERS)} (hpo_train.py:215-219)")
        if loss not in LOSSES:
            raise ValueError(f"loss must be one of {sorted(LOSSES)} (hpo_train.py:221-226)")
        self.lib = _lib.load()
        self.depth, self.channel_width = depth, channel_width
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.max_batch = int(max_batch)
        self.trainable, self.loss_name, self.dropout = bool(trainable), loss, float(dropout)
        self._seed = int(seed)
        self.iterations = 0
        cfg = _lib.CsCnnCfg(depth=depth, channels=channel_width, kernel=kernel_width, seq=60, c_in=6, c_out=10, n_lin=2,
                            max_batch=self.max_batch, device=self.device_index, flags=int(bool(tile128)), train=int(self.trainable),
                            optimizer=OPTIMIZERS[optimizer], loss=LOSSES[loss], reserved=0, dropout=self.dropout,
                            beta1=beta_1, beta2=beta_2, eps=epsilon, seed=int(seed))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cs_cnn_create(C.byref(self._h), C.byref(cfg)))
        self._n_params = int(self.lib.cs_cnn_num_params(self._h))
        self._shapes = _shapes(depth, channel_width)
        self._loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        # running sums of the reference's two remaining compiled metrics (hpo_train.py:83-111, 231): [CRPS score terms, argmax matches]
        self._metrics = torch.zeros(2, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.cs_cnn_set_metrics_buffer(self._h, _ptr(self._metrics)))
        self._grad_tensor = None
        if self.trainable:
            self.gradient_tensor()                   # bind the torch-owned gradient buffer before the first step
        if init_seed is not None:
            self.set_weights(glorot_uniform_cnn(depth, channel_width, init_seed))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.cs_cnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def count_params(self) -> int:
        return self._n_params

    def set_weights(self, weights: List[np.ndarray]):
        """model.set_weights: Keras order, Conv1D kernels (k, c_in, c_out)."""
        if len(weights) != len(self._shapes):
            raise ValueError(f"expected {len(self._shapes)} arrays, got {len(weights)}")
        for w, s in zip(weights, self._shapes):
            if tuple(np.shape(w)) != tuple(s):
                raise ValueError(f"weight shape {np.shape(w)} does not match {s}")
        flat = np.ascontiguousarray(np.concatenate([np.asarray(w, np.float32).ravel() for w in weights]))
        _lib.check(self.lib.cs_cnn_set_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size, self._stream()))

    def predict(self, x, flat_output: bool = False, as_numpy: bool = True):
        """model.predict.  x: (N,60,6) like train_input_cnn.npy, or (N,124) flat rows (the reshape of
        data_utils.reshape_input_for_cnn then happens on the GPU).  Returns (N,60,10), or (N,128) in the
        layout of data_utils.reshape_target_from_cnn when flat_output."""
        import torch
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if x.ndim == 3 and x.shape[1:] == (60, 6):
            layout3d = 1
        elif x.ndim == 2 and x.shape[1] == 124:
            layout3d = 0
        else:
            raise ValueError(f"expected (N,60,6) or (N,124), got {tuple(x.shape)}")
        n = x.shape[0]
        out = torch.empty((n, 128) if flat_output else (n, 60, 10), dtype=torch.float32, device=self.device)
        row = x[0].numel()
        orow = out[0].numel()
        for lo in range(0, n, self.max_batch):
            m = min(self.max_batch, n - lo)
            xp = C.c_void_p(x.data_ptr() + lo * row * 4)
            op = C.c_void_p(out.data_ptr() + lo * orow * 4)
            _lib.check(self.lib.cs_cnn_forward(self._h, xp, layout3d, m, None if flat_output else op,
                                               op if flat_output else None, self._stream()))
        return out.cpu().numpy() if as_numpy else out


    # ---- weights / optimiser state
    def _split(self, flat):
        out, at = [], 0
        for sh in self._shapes:
            k = int(np.prod(sh))
            out.append(flat[at:at + k].reshape(sh).copy())
            at += k
        return out

    def get_weights(self) -> List[np.ndarray]:
        flat = np.empty(self._n_params, np.float32)
        _lib.check(self.lib.cs_cnn_get_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size, self._stream()))
        return self._split(flat)

    def get_optimizer_state(self):
        m, v = np.empty(self._n_params, np.float32), np.empty(self._n_params, np.float32)
        it = C.c_int64(0)
        _lib.check(self.lib.cs_cnn_get_opt_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                                 m.size, C.byref(it), self._stream()))
        return self._split(m), self._split(v), int(it.value)

    def set_optimizer_state(self, m, v, iterations):
        fm = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float32).ravel() for a in m]))
        fv = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float32).ravel() for a in v]))
        _lib.check(self.lib.cs_cnn_set_opt_state(self._h, fm.ctypes.data_as(C.c_void_p), fv.ctypes.data_as(C.c_void_p),
                                                 fm.size, int(iterations), self._stream()))
        self.iterations = int(iterations)

    def save_weights(self, path: str):
        """ModelCheckpoint(save_weights_only=False) counterpart: weights + optimiser slots in one .npz."""
        d = {f"w{i}": w for i, w in enumerate(self.get_weights())}
        if self.trainable:
            m, v, it = self.get_optimizer_state()
            d.update({f"m{i}": a for i, a in enumerate(m)})
            d.update({f"v{i}": a for i, a in enumerate(v)})
            d["iterations"] = np.int64(it)
        final = path if path.endswith(".npz") else path + ".npz"       # np.savez appends .npz itself
        tmp = final + ".tmp.npz"
        np.savez(tmp, **d)
        os.replace(tmp, final)                                          # a reader never sees a half-written checkpoint

    def load_weights(self, path: str, with_optimizer: bool = True):
        z = np.load(path if path.endswith(".npz") else path + ".npz")
        k = len(self._shapes)
        self.set_weights([z[f"w{i}"] for i in range(k)])
        if with_optimizer and self.trainable and "iterations" in z:
            self.set_optimizer_state([z[f"m{i}"] for i in range(k)], [z[f"v{i}"] for i in range(k)], int(z["iterations"]))

    # ---- training
    def _to_device(self, a, flat_cols, shape3):
        import torch
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if a.ndim == 3 and tuple(a.shape[1:]) == shape3:
            return a, 1
        if a.ndim == 2 and a.shape[1] == flat_cols:
            return a, 0
        raise ValueError(f"expected (N,{shape3[0]},{shape3[1]}) or (N,{flat_cols}), got {tuple(a.shape)}")

    def _losses(self, sums, n_cols, metrics=None):
        """Loss sums (+ optionally the metric sums of `self._metrics`) of `n_cols` columns -> the values Keras logs for
        compile(metrics=["mse", "mae", "accuracy", mse_adjusted, mae_adjusted, continuous_ranked_probability_score]), hpo_train.py:231."""
        s = np.asarray(sums, np.float64)
        dp, ds = n_cols * 60 * 2, n_cols * 60 * 8
        mae = s[0] / dp * (120 / 128) + s[1] / ds * (8 / 128)
        mse = s[2] / dp * (120 / 128) + s[3] / ds * (8 / 128)
        out = {"loss": float(mae if LOSSES[self.loss_name] == 0 else mse), "mae_adjusted": float(mae), "mse_adjusted": float(mse),
               "mae": float((s[0] + s[1]) / (dp + ds)), "mse": float((s[2] + s[3]) / (dp + ds))}
        if metrics is not None:
            q = np.asarray(metrics, np.float64)
            out["continuous_ranked_probability_score"] = float(q[0] / (n_cols * 60))
            out["accuracy"] = float(q[1] / (n_cols * 60))
        return out

    def loss_grads(self, x, y, row_idx=None, n=None, loss=None, normalise=False, x3d=None, y3d=None):
        """Training-mode forward (dropout on) + backward of one batch; gradients (unscaled sums) land in
        `gradient_tensor()`.  x, y are device tensors (see `_to_device`), row_idx an int64 device tensor."""
        if x3d is None:
            x, x3d = self._to_device(x, 124, (60, 6))
        if y3d is None:
            y, y3d = self._to_device(y, 128, (60, 10))
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        loss = self._loss if loss is None else loss
        _lib.check(self.lib.cs_cnn_loss_grads(self._h, _ptr(x), x3d, _ptr(y), y3d, _ptr(row_idx), n, _ptr(loss), self._stream()))
        return loss

    def apply_gradients(self, lr: float, grad_scale: float):
        _lib.check(self.lib.cs_cnn_apply(self._h, float(lr), float(grad_scale), self._stream()))
        self.iterations += 1

    def train_on_batch(self, x, y, lr: float, row_idx=None, n=None, loss=None, x3d=None, y3d=None):
        """One Model.train_step; returns the device tensor of the four loss sums."""
        if x3d is None:
            x, x3d = self._to_device(x, 124, (60, 6))
        if y3d is None:
            y, y3d = self._to_device(y, 128, (60, 10))
        n = int(n if n is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        loss = self._loss if loss is None else loss
        _lib.check(self.lib.cs_cnn_train_step(self._h, _ptr(x), x3d, _ptr(y), y3d, _ptr(row_idx), n, float(lr), _ptr(loss),
                                              self._stream()))
        self.iterations += 1
        return loss

    def gradient_tensor(self):
        """Flat float32 gradient buffer (Keras order) as a torch tensor bound into the engine - the payload
        of the one-per-step RCCL all-reduce."""
        if self._grad_tensor is None:
            import torch
            self._grad_tensor = torch.zeros(self._n_params, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.cs_cnn_set_grad_buffer(self._h, _ptr(self._grad_tensor), self._n_params))
        return self._grad_tensor

    def bind_gradient_tensor(self, tensor):
        """Make `tensor` the engine's flat gradient buffer (the exchange buffer of the one-shot all-reduce, dp.py: IpcComm)."""
        import torch
        if tensor.numel() < self._n_params or tensor.dtype != torch.float32 or not tensor.is_cuda:
            raise ValueError("gradient buffer must be a float32 device tensor of at least %d elements" % self._n_params)
        tensor.zero_()
        self._grad_tensor = tensor[:self._n_params]
        _lib.check(self.lib.cs_cnn_set_grad_buffer(self._h, _ptr(self._grad_tensor), self._n_params))

    def get_gradients(self, grad_scale: float = 1.0) -> List[np.ndarray]:
        return self._split(self.gradient_tensor().detach().cpu().numpy() * np.float32(grad_scale))

    def evaluate(self, x, y, batch_size: Optional[int] = None):
        """model.evaluate: loss (the compiled one) plus every compiled metric: mae / mse and their adjusted forms, accuracy and
        continuous_ranked_probability_score (hpo_train.py:227-231)."""
        import torch
        x, x3d = self._to_device(x, 124, (60, 6))
        y, y3d = self._to_device(y, 128, (60, 10))
        bs = min(batch_size or self.max_batch, self.max_batch)
        tot = torch.zeros(4, dtype=torch.float32, device=self.device)
        kept = self._metrics.clone()                     # an evaluation inside fit() must not disturb the epoch's running sums
        self._metrics.zero_()
        for i, lo in enumerate(range(0, x.shape[0], bs)):
            hi = min(lo + bs, x.shape[0])
            _lib.check(self.lib.cs_cnn_evaluate(self._h, _ptr(x[lo:hi]), x3d, _ptr(y[lo:hi]), y3d, None, hi - lo, _ptr(tot),
                                                int(i > 0), self._stream()))
        out = self._losses(tot.cpu().numpy(), x.shape[0], self._metrics.cpu().numpy())
        self._metrics.copy_(kept)
        return out

    def fit(self, x, y, batch_size: int = 512, epochs: int = 15, validation_data=None, learning_rate=None,
            shuffle: bool = True, seed: int = 0, steps_per_epoch: Optional[int] = None, checkpoint: Optional[str] = None,
            early_stopping_patience: Optional[int] = 10, distributed: bool = False, verbose: int = 0):
        """model.fit of hpo_train.py:355-368 on an HBM-resident split: batches of `batch_size` columns drawn from a
        per-epoch device permutation (the reference streams a Python generator through shuffle(2000).batch(512,
        drop_remainder=True)), Adam/SGD with the cyclical schedule, validation pass, EarlyStopping('val_loss',
        patience) and a per-epoch ModelCheckpoint (`checkpoint` may contain '{epoch}').  With `distributed`,
        `batch_size` is the global batch dealt round-robin to the ranks (see climsim_amd.dp)."""
        import torch
        if not self.trainable:
            raise _lib.EngineError("create the emulator with trainable=True to fit")
        x, x3d = self._to_device(x, 124, (60, 6))
        y, y3d = self._to_device(y, 128, (60, 10))
        sched = learning_rate or cnn_learning_rate(self.depth)
        if not callable(sched):
            sched = ConstantLearningRate(float(sched))
        from .dp import DataParallel, shard_of_batch
        dist = None
        if distributed:
            import torch.distributed as dist
        dp = DataParallel(self, dist, 60)
        rank, world = dp.rank, dp.world
        if distributed:
            dp.broadcast_weights()
            # every rank draws its own dropout masks (same weights, different rows AND different masks)
            _lib.check(self.lib.cs_cnn_set_seed(self._h, C.c_uint64((self._seed + 1000003 * rank) & 0xFFFFFFFFFFFFFFFF)))
        if batch_size % world or batch_size // world > self.max_batch:
            raise ValueError("global batch must be divisible by the world size and fit max_batch per GPU")
        n = x.shape[0]
        steps = steps_per_epoch or (n // batch_size)
        if steps < 1:
            raise ValueError("dataset smaller than one batch")
        gen = torch.Generator(device=self.device)
        keys = ["loss", "mse", "mae", "accuracy", "mse_adjusted", "mae_adjusted", "continuous_ranked_probability_score"]   # Keras' order
        history = {k: [] for k in keys + ["lr"]}
        if validation_data is not None:
            history.update({"val_" + k: [] for k in keys})
        epoch_sum = torch.zeros(4, dtype=torch.float32, device=self.device)
        step_loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        best, wait = math.inf, 0
        self.stop_training = False
        try:
            for epoch in range(epochs):
                gen.manual_seed(seed + epoch)
                perm = torch.randperm(n, device=self.device, generator=gen) if shuffle else torch.arange(n, device=self.device)
                epoch_sum.zero_()
                self._metrics.zero_()
                lr = sched(self.iterations)
                for s in range(steps):
                    lr = sched(self.iterations)
                    if distributed:
                        idx = shard_of_batch(perm, s, batch_size, rank, world)
                        self.loss_grads(x, y, row_idx=idx, loss=step_loss, x3d=x3d, y3d=y3d)
                        dp.all_reduce_grads()
                        self.apply_gradients(lr, 1.0 / (60 * batch_size))
                    else:
                        idx = shard_of_batch(perm, s, batch_size, 0, 1)
                        self.train_on_batch(x, y, lr, row_idx=idx, loss=step_loss, x3d=x3d, y3d=y3d)
                    epoch_sum += step_loss
                if distributed:
                    dist.all_reduce(epoch_sum)
                    dist.all_reduce(self._metrics)
                row = self._losses(epoch_sum.cpu().numpy(), batch_size * steps, self._metrics.cpu().numpy())
                row["lr"] = float(lr)
                if validation_data is not None:
                    ev = self.evaluate(validation_data[0], validation_data[1])
                    row.update({"val_" + k: ev[k] for k in keys})
                for k, v in row.items():
                    history[k].append(v)
                if verbose and rank == 0:
                    print(f"epoch {epoch + 1}/{epochs} " + " ".join(f"{k}={v:.6g}" for k, v in row.items()), flush=True)
                monitor = row.get("val_loss", row["loss"])
                if not math.isfinite(monitor):
                    raise FloatingPointError(f"non-finite loss at epoch {epoch}")
                if rank == 0 and checkpoint:
                    self.save_weights(checkpoint.format(epoch=epoch + 1))
                if monitor < best:
                    best, wait = monitor, 0
                else:
                    wait += 1
                    if early_stopping_patience is not None and wait >= early_stopping_patience:
                        self.stop_training = True
                        break
        finally:
            dp.close()                     # the RCCL communicator never outlives the call, also on an exception
        return history
