"""Device-side evaluation metrics: `data_utils.set_pressure_grid` + `output_weighting` + `calc_MAE / calc_RMSE / calc_R2 /
calc_bias` + the tables of `create_metrics_df` (climsim_utils/data_utils.py:1037-1086, 1112-1362, 1432-1497, 1526-1607)
on predictions that already live in HBM - the held-out score of every epoch without a host round trip.
One kernel (`cs_metrics_columns`, csrc/metrics.h), float64 accumulation; the per-feature constants of the weighting
(1/out_scale, layer thickness coefficients over g, energy-unit factor) are folded into two vectors on the host."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .data_utils import _values


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class GpuMetrics:
    METRICS = ("MAE", "RMSE", "R2", "bias")

    def __init__(self, du, device=None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.EngineError("GpuMetrics needs a ROCm GPU (no CPU fallback; use data_utils on the host)")
        if du.full_vars:
            raise ValueError("the wind-speed weighting of the v2 targets is not implemented on the device")
        self.lib = _lib.load()
        self.du = du
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        p0 = float(_values(du.grid_info["P0"]))
        hyai, hybi = np.asarray(_values(du.grid_info["hyai"]), np.float64), np.asarray(_values(du.grid_info["hybi"]), np.float64)
        da, db = (hyai[1:61] - hyai[0:60]) * p0, hybi[1:61] - hybi[0:60]
        wa, wb = [], []
        for v in du.target_vars:
            ln = du.var_lens[v]
            conv = float(du.target_energy_conv[v])
            sc = np.broadcast_to(np.asarray(_values(du.output_scale[v]), np.float64), (ln,)) if du.normalize else np.ones(ln)
            if ln > 1:
                wa.append(conv / sc * da / du.grav)
                wb.append(conv / sc * db / du.grav)
            else:
                wa.append(conv / sc)
                wb.append(np.zeros(1))
        f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(self.device)  # noqa: E731
        self._wa, self._wb = f64(np.concatenate(wa)), f64(np.concatenate(wb))
        self._area = f64(du.area_wgt)
        self.n_out = int(self._wa.numel())
        self.ncol = int(du.num_latlon)
        if du.normalize:
            self._ps_mul = float(_values(du.input_max["state_ps"]) - _values(du.input_min["state_ps"]))
            self._ps_add = float(_values(du.input_mean["state_ps"]))
        else:
            self._ps_mul, self._ps_add = 1.0, 0.0

    def column_stats(self, preds, target, inputs):
        """(ncol, n_out, 4) float64 device tensor of MAE, RMSE, R2, bias per grid column and output (avg_grid=False).
        preds/target: (N, n_out) float32, inputs: (N, n_in) normalised rows (for the surface pressure); numpy or device."""
        import torch
        dev = lambda a: (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(self.device)  # noqa: E731
        p, t, x = dev(preds).float().contiguous(), dev(target).float().contiguous(), dev(inputs)
        n = p.shape[0]
        if tuple(t.shape) != (n, self.n_out) or tuple(p.shape) != (n, self.n_out) or x.shape[0] != n or n % self.ncol:
            raise ValueError(f"expected (T*{self.ncol}, {self.n_out}) predictions and targets and matching inputs")
        out = torch.empty((self.ncol, self.n_out, 6), dtype=torch.float64, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        import os
        if (x.dtype == torch.float32 and x.is_contiguous() and self.n_out % 4 == 0 and p.data_ptr() % 16 == 0 and t.data_ptr() % 16 == 0
                and os.environ.get("CS_METRICS_V5", "1") != "0" and os.environ.get("CS_METRICS_V4", "1") != "0"):
            # round 5: the kernel reads the surface pressure from the input rows itself (no gather / conversion passes over the rows)
            _lib.check(self.lib.cs_metrics_columns_x(_ptr(p), _ptr(t), n // self.ncol, self.ncol, self.n_out, _ptr(x), int(x.shape[1]),
                                                     int(self.du.ps_index), float(self._ps_mul), float(self._ps_add), _ptr(self._wa),
                                                     _ptr(self._wb), _ptr(self._area), _ptr(out), st))
            return out[:, :, :4]
        ps = (x[:, self.du.ps_index].double() * self._ps_mul + self._ps_add).contiguous()
        _lib.check(self.lib.cs_metrics_columns(_ptr(p), _ptr(t), n // self.ncol, self.ncol, self.n_out, _ptr(ps), _ptr(self._wa),
                                               _ptr(self._wb), _ptr(self._area), _ptr(out), st))
        return out[:, :, :4]

    def metrics_tables(self, preds, target, inputs):
        """(df_var, df_idx) like data_utils.create_metrics_df: per-variable means and per-output values, grid-averaged."""
        import pandas as pd
        stats = self.column_stats(preds, target, inputs).mean(dim=0).cpu().numpy()          # (n_out, 4): mean over the grid
        df_idx = pd.DataFrame(stats, columns=list(self.METRICS), index=range(self.n_out))
        df_idx.index.name = "output_idx"
        rows, at = [], 0
        for v in self.du.target_vars:
            ln = self.du.var_lens[v]
            rows.append(stats[at:at + ln].mean(axis=0))
            at += ln
        df_var = pd.DataFrame(np.asarray(rows), columns=list(self.METRICS), index=list(self.du.target_vars))
        df_var.index.name = "variable"
        return df_var, df_idx
