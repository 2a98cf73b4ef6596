#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference and /opt/conda/bin/h5dump); the outputs
(`*.npz`) are committed, the reference source never is.  What is pinned:

  G0  static assets  : the four v1/v2 normalisation datasets and the low-res grid file, as data
  G1  save_norm      : input_sub / input_div / out_scale for v1 and v2   (data_utils.py:954-988)
  G2  output_weighting (+ set_pressure_grid) on deterministic inputs     (:1037-1086, :1112-1362)
  G3  calc_MAE / RMSE / R2 / bias + create_metrics_df tables             (:1432-1497, :1526-1607)
  G4  calc_CRPS on deterministic sample predictions                      (:1499-1524)
  G5  CNN reshape trio                                                   (:1692-1761)
  G6  output_weighting(just_weights=True)

The reference module imports xarray / tensorflow / netCDF4 / h5py at the top although its numpy
paths never touch them; they are absent here, so empty stub modules are installed first.
Inputs are produced by `golden_inputs.py` (exact integer hashing -> float), so the tests can
rebuild them bit-for-bit without storing them.
"""
import os
import re
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from climsim_amd.assets import AssetSet, read_cdf5, load_grid_info, load_npz_assets  # noqa: E402
from golden_inputs import make_metric_inputs, make_crps_inputs, make_cnn_inputs, subsample  # noqa: E402

H5DUMP = "/opt/conda/bin/h5dump"


def dump_h5(path):
    """All datasets of a (scalar / 1-D f64) HDF5 file via h5dump text output."""
    names = subprocess.run([H5DUMP, "-n", path], check=True, capture_output=True, text=True).stdout
    out = {}
    for nm in re.findall(r"dataset\s+/(\S+)", names):
        txt = subprocess.run([H5DUMP, "-y", "-w", "100000", "-m", "%.17g", "-d", nm, path],
                             check=True, capture_output=True, text=True).stdout
        head, data = txt.split("DATA {", 1)
        data = data.split("}", 1)[0]
        vals = [float(t) for t in re.findall(r"[-+0-9.eEnaif]+", data.replace(",", " "))]
        out[nm] = np.float64(vals[0]) if "SCALAR" in head else np.asarray(vals, dtype=np.float64)
    return out


def main():
    # ---- G0: assets --------------------------------------------------------------------------
    norm = {}
    for key, rel in [("input_mean", "inputs/input_mean.nc"), ("input_max", "inputs/input_max.nc"),
                     ("input_min", "inputs/input_min.nc"), ("output_scale", "outputs/output_scale.nc")]:
        for var, val in dump_h5(f"{REF}/preprocessing/normalizations/{rel}").items():
            norm[f"{key}/{var}"] = val
    np.savez_compressed(f"{HERE}/norm_lowres.npz", **norm)
    grid_raw = read_cdf5(f"{REF}/grid_info/ClimSim_low-res_grid-info.nc")
    grid_raw.pop("__dims__")
    np.savez_compressed(f"{HERE}/grid_lowres.npz", **grid_raw)

    # ---- import the reference with stub modules ------------------------------------------------
    for mod in ("xarray", "tensorflow", "netCDF4", "h5py"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.path.insert(0, REF)
    from climsim_utils.data_utils import data_utils as ref_data_utils  # noqa: E402

    def build():
        grid = load_grid_info(f"{HERE}/grid_lowres.npz")
        sets = [load_npz_assets(f"{HERE}/norm_lowres.npz", k)
                for k in ("input_mean", "input_max", "input_min", "output_scale")]
        return ref_data_utils(grid, *sets, ml_backend="pytorch")

    gold = {}

    def big(key, arr):
        """Large arrays are pinned by a strided subsample plus first/second moments."""
        arr = np.asarray(arr, dtype=np.float64)
        gold[key + "@sub"] = subsample(arr)
        gold[key + "@mom"] = np.array([arr.sum(), (arr * arr).sum(), np.abs(arr).max()])
        gold[key + "@shape"] = np.array(arr.shape)

    # ---- G1 ------------------------------------------------------------------------------------
    d = build()
    d.set_to_v1_vars()
    s, v, o = d.save_norm()
    gold["g1_v1_input_sub"], gold["g1_v1_input_div"], gold["g1_v1_out_scale"] = (
        np.asarray(s, dtype=np.float64), np.asarray(v, dtype=np.float64), np.asarray(o, dtype=np.float64))
    d2 = build()
    d2.set_to_v2_vars()
    with np.errstate(all="ignore"):
        s, v, o = d2.save_norm()
    gold["g1_v2_input_sub"], gold["g1_v2_input_div"], gold["g1_v2_out_scale"] = (
        np.asarray(s, dtype=np.float64), np.asarray(v, dtype=np.float64), np.asarray(o, dtype=np.float64))

    # ---- G2 / G3 / G6 ----------------------------------------------------------------------------
    T = 4
    x, y, p = make_metric_inputs(T)
    d.input_scoring, d.target_scoring = x, y
    d.set_pressure_grid("scoring")
    big("g2_dp", d.dp_scoring)
    d.model_names = ["MLP"]
    d.preds_scoring = {"MLP": p}
    d.reweight_target("scoring")
    d.reweight_preds("scoring")
    for var in d.target_vars:
        big(f"g2_target_weighted/{var}", d.target_weighted_scoring[var])
        big(f"g2_preds_weighted/{var}", d.preds_weighted_scoring["MLP"][var])
    d.metrics_names = ["MAE", "RMSE", "R2", "bias"]
    with np.errstate(all="ignore"):
        d.create_metrics_df("scoring")
        for var in d.target_vars:
            for m in d.metrics_names:
                fn = d.metrics_dict[m]
                gold[f"g3_{m}/{var}"] = np.asarray(fn(d.preds_weighted_scoring["MLP"][var],
                                                      d.target_weighted_scoring[var]))
                gold[f"g3_{m}_grid/{var}"] = np.asarray(fn(d.preds_weighted_scoring["MLP"][var],
                                                           d.target_weighted_scoring[var], avg_grid=False))
    gold["g3_df_var"] = d.metrics_var_scoring["MLP"].to_numpy(dtype=np.float64)
    gold["g3_df_idx"] = d.metrics_idx_scoring["MLP"].to_numpy(dtype=np.float64)
    big("g6_weights", d.output_weighting(y, "scoring", just_weights=True))

    # ---- G4 ------------------------------------------------------------------------------------
    sp3, t3, sp2, t2 = make_crps_inputs()
    gold["g4_crps_3d"] = d.calc_CRPS(sp3, t3)
    gold["g4_crps_3d_grid"] = d.calc_CRPS(sp3, t3, avg_grid=False)
    gold["g4_crps_2d"] = d.calc_CRPS(sp2, t2)

    # ---- G5 ------------------------------------------------------------------------------------
    xi, yi = make_cnn_inputs()
    xc = ref_data_utils.reshape_input_for_cnn(xi)
    yc = ref_data_utils.reshape_target_for_cnn(yi)
    back = ref_data_utils.reshape_target_from_cnn(yc)
    gold["g5_input_cnn"], gold["g5_target_cnn"], gold["g5_target_back"] = xc, yc, back
    # a non-constant scalar channel exercises the level-mean of the inverse
    yc2 = yc + np.linspace(0, 1, 60, dtype=yc.dtype)[None, :, None]
    gold["g5_target_back_mean"] = ref_data_utils.reshape_target_from_cnn(yc2)

    # attributes that callers read
    gold["attr_area_wgt"] = d.area_wgt
    gold["attr_lats"] = d.lats
    gold["attr_lons"] = d.lons
    np.savez_compressed(f"{HERE}/data_utils_golden.npz", **gold)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
