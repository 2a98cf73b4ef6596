"""Host side of the level-axis CNN engine (prediction path): counterpart of the Keras model built by
`CNNHyperModel.build` (baseline_models/CNN/training/hpo_train.py:124-236).  Same conventions as
`climsim_amd.mlp`: arithmetic in libclimsim_hip.so, torch only for device memory and streams, no CPU
fallback.  Training (mae_adjusted, Adam + cyclical LR, dropout) is the next step - see DESIGN.md."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _lib


def _shapes(depth, channels, c_in=6, c_out=10, n_lin=2, k=3):
    sh = []
    for b in range(depth):
        cin = c_in if b == 0 else channels
        sh += [(k, cin, channels), (channels,), (k, channels, channels), (channels,), (1, cin, channels), (channels,)]
    sh += [(1, channels, c_out), (c_out,), (c_out, n_lin), (n_lin,), (c_out, c_out - n_lin), (c_out - n_lin,)]
    return sh


class CNNEmulator:
    """(B,60,6) -> (B,60,10) ResNet-style 1-D CNN over the level axis on one MI355X."""

    def __init__(self, depth: int = 12, channel_width: int = 406, kernel_width: int = 3, max_batch: int = 512,
                 device: Optional[int] = None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.EngineError("CNNEmulator needs a ROCm GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.depth, self.channel_width = depth, channel_width
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.max_batch = int(max_batch)
        cfg = _lib.CsCnnCfg(depth=depth, channels=channel_width, kernel=kernel_width, seq=60, c_in=6, c_out=10, n_lin=2,
                            max_batch=self.max_batch, device=self.device_index, flags=0)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cs_cnn_create(C.byref(self._h), C.byref(cfg)))
        self._n_params = int(self.lib.cs_cnn_num_params(self._h))
        self._shapes = _shapes(depth, channel_width)

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
