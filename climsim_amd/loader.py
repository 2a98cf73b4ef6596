"""Device-side column loader: raw E3SM-MMF timestep files -> normalised float32 training rows in HBM.

Counterpart of `data_utils.load_ncdata_with_generator` + `save_as_npy` (climsim_utils/data_utils.py:791-944):
the host only reads the files (classic netCDF through `climsim_amd.assets`) and copies the raw fields, feature-major
as they are stored, to the GPU; tendencies, normalisation, inf/nan -> 0, stacking and the float32 cast run in ONE
HBM-bound kernel (`cs_loader_stack`, csrc/loader.h) with float64 arithmetic, so the rows are bit-identical to the
`.npy` files the reference writes.  Variable sets whose inputs are raw file fields (v1; v2 without the derived
humidity / partition inputs) are supported; derived inputs raise."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .data_utils import _open_columns


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class GpuColumnLoader:
    def __init__(self, du, device=None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.EngineError("GpuColumnLoader needs a ROCm GPU (no CPU fallback; use data_utils on the host)")
        self.lib = _lib.load()
        self.du = du
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.in_rows = [(v, l) for v in du.input_vars for l in range(du.var_lens[v])]
        self.out_rows = [(v, l) for v in du.target_vars for l in range(du.var_lens[v])]
        self.n_in, self.n_out = len(self.in_rows), len(self.out_rows)
        index = {vl: i for i, vl in enumerate(self.in_rows)}
        tend = []
        for v, l in self.out_rows:
            if v.startswith("ptend_"):
                state = "state_" + v[len("ptend_"):]
                if (state, l) not in index:
                    raise ValueError(f"tendency target {v} needs {state} among the input variables")
                tend.append(index[(state, l)])
            else:
                tend.append(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            sub, div, scale = du.save_norm()
        if not du.normalize:
            sub, div, scale = np.zeros(self.n_in), np.ones(self.n_in), np.ones(self.n_out)
        f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(self.device)  # noqa: E731
        self._sub, self._div, self._scale = f64(sub), f64(div), f64(scale)
        self._tend = torch.from_numpy(np.asarray(tend, np.int32)).to(self.device)

    # ---- host side: one file pair -> feature-major raw blocks
    def read_raw(self, input_file: str):
        du = self.du
        mli = _open_columns(input_file)
        mlo = _open_columns(input_file.replace(f".{du.input_abbrev}.", f".{du.output_abbrev}."))
        for v in du.input_vars:
            if v not in mli:
                raise ValueError(f"input {v} is derived on the host by data_utils; not available in the device loader")
        a = np.concatenate([np.asarray(mli[v], np.float64).reshape(du.var_lens[v], -1) for v in du.input_vars])
        rows = []
        for v in du.target_vars:
            src = "state_" + v[len("ptend_"):] if v.startswith("ptend_") else v
            rows.append(np.asarray(mlo[src], np.float64).reshape(du.var_lens[v], -1))
        return a, np.concatenate(rows)

    def stack_raw(self, mli_raw, mlo_raw=None, want_x=True, want_y=True, extra_rows: int = 0):
        """mli_raw (T, n_in, ncol), mlo_raw (T, n_out, ncol): numpy or device tensors, float64 or float32.
        Returns (x (T*ncol, n_in), y (T*ncol, n_out)) float32 device tensors.  `extra_rows`: the tensors are allocated that many rows
        longer (uninitialised tail: the streamed trainer appends the rows the previous chunk's last batch left over)."""
        import torch
        def dev(a):
            if a is None:
                return None
            if isinstance(a, np.ndarray):
                a = torch.from_numpy(np.ascontiguousarray(a))
            return a.to(self.device).contiguous()
        a, b = dev(mli_raw), dev(mlo_raw)
        if a.dtype not in (torch.float64, torch.float32) or (b is not None and b.dtype != a.dtype):
            raise ValueError("raw fields must be float64 or float32 (both the same)")
        T, fin, ncol = a.shape
        if fin != self.n_in or (b is not None and tuple(b.shape) != (T, self.n_out, ncol)):
            raise ValueError(f"expected (T,{self.n_in},ncol) and (T,{self.n_out},ncol)")
        want_y = want_y and b is not None
        x = torch.empty((T * ncol + extra_rows, self.n_in), dtype=torch.float32, device=self.device) if want_x else None
        y = torch.empty((T * ncol + extra_rows, self.n_out), dtype=torch.float32, device=self.device) if want_y else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for lo in range(0, T, 32768):                         # grid.y limit
            hi = min(T, lo + 32768)
            _lib.check(self.lib.cs_loader_stack(_ptr(a[lo:hi]), _ptr(b[lo:hi]) if b is not None else None, int(a.dtype == torch.float64),
                                                hi - lo, ncol, self.n_in, _ptr(self._sub), _ptr(self._div), self.n_out,
                                                _ptr(self._tend), _ptr(self._scale),
                                                _ptr(x[lo * ncol:]) if want_x else None, _ptr(y[lo * ncol:]) if want_y else None, st))
        return x, y

    def stack_raw_sliced(self, mli_raw, mlo_raw, extra_rows: int = 0):
        """The rows of `stack_raw` produced a few timesteps at a time: returns (x, y, run) with x, y allocated (uninitialised) and
        `run(t_lo, t_hi)` launching the loader for timesteps [t_lo, t_hi) on the CURRENT stream into their rows.  The streamed trainer
        (stream.py, loader_on="gaps") spreads a chunk's loader over the steps of the chunk in front.  Device tensors only."""
        import torch
        a, b = mli_raw, mlo_raw
        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
            raise ValueError("stack_raw_sliced takes device tensors")
        a, b = a.contiguous(), b.contiguous()
        if a.dtype not in (torch.float64, torch.float32) or b.dtype != a.dtype:
            raise ValueError("raw fields must be float64 or float32 (both the same)")
        T, fin, ncol = a.shape
        if fin != self.n_in or tuple(b.shape) != (T, self.n_out, ncol):
            raise ValueError(f"expected (T,{self.n_in},ncol) and (T,{self.n_out},ncol)")
        x = torch.empty((T * ncol + extra_rows, self.n_in), dtype=torch.float32, device=self.device)
        y = torch.empty((T * ncol + extra_rows, self.n_out), dtype=torch.float32, device=self.device)
        f64 = int(a.dtype == torch.float64)

        def run(t_lo: int, t_hi: int):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.lib.cs_loader_stack(_ptr(a[t_lo:t_hi]), _ptr(b[t_lo:t_hi]), f64, t_hi - t_lo, ncol, self.n_in, _ptr(self._sub),
                                                _ptr(self._div), self.n_out, _ptr(self._tend), _ptr(self._scale),
                                                _ptr(x[t_lo * ncol:]), _ptr(y[t_lo * ncol:]), st))
        run.keep = (a, b)
        return x, y, run

    def load_files(self, files: Sequence[str]):
        """The rows `save_as_npy` would write for these mli files, as float32 device tensors."""
        raws = [self.read_raw(f) for f in files]
        return self.stack_raw(np.stack([r[0] for r in raws]), np.stack([r[1] for r in raws]))

    def load_split(self, data_split: str):
        return self.load_files(self.du.get_filelist(data_split))
