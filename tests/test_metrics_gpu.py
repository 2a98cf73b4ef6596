"""Device metrics (cs_metrics_columns) against the golden-pinned host pipeline of climsim_amd.data_utils
(set_pressure_grid -> output_weighting -> calc_* -> create_metrics_df).  float64 on both sides; tolerance 1e-9 relative
(different summation order and the folded weighting constants)."""
import copy

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from golden_inputs import make_metric_inputs  # noqa: E402


@pytest.mark.parametrize("T", [9, 200])         # 200 time steps: three slices of the time axis, combined with float64 atomics
def test_device_metrics_match_host_pipeline(lowres_assets, T):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd import build
    build.build()
    from climsim_amd.data_utils import data_utils
    from climsim_amd.metrics import GpuMetrics
    grid, *sets = lowres_assets
    d = data_utils(copy.copy(grid), *sets)
    d.set_to_v1_vars()
    x, y, p = make_metric_inputs(T)
    d.input_scoring, d.target_scoring = x, y
    d.set_pressure_grid("scoring")
    d.model_names = ["m"]
    d.preds_scoring = {"m": p}
    d.reweight_target("scoring")
    d.reweight_preds("scoring")
    d.metrics_names = ["MAE", "RMSE", "R2", "bias"]
    with np.errstate(all="ignore"):
        d.create_metrics_df("scoring")
    gm = GpuMetrics(d)
    stats = gm.column_stats(p, y, x).cpu().numpy()                 # (384, 128, 4)
    # round 5: float32 input rows take cs_metrics_columns_x (surface pressure read inside the kernel); float64 rows the entry that is
    # handed a prepared float64 pressure array - the same float64 arithmetic either way
    assert x.dtype == np.float32
    via_ps = gm.column_stats(p, y, x.astype(np.float64)).cpu().numpy()
    both = np.isfinite(stats) & np.isfinite(via_ps)
    np.testing.assert_allclose(stats[both], via_ps[both], rtol=1e-10, atol=1e-13)      # (float64 atomics arrive in any order)
    assert np.array_equal(np.isfinite(stats), np.isfinite(via_ps))
    at = 0
    for v in d.target_vars:
        ln = d.var_lens[v]
        pw, tw = d.preds_weighted_scoring["m"][v], d.target_weighted_scoring[v]
        for k, name in enumerate(d.metrics_names):
            with np.errstate(all="ignore"):
                ref = d.metrics_dict[name](pw, tw, avg_grid=False)                 # (384[, 60])
            got = stats[:, at:at + ln, k] if ln > 1 else stats[:, at, k]
            scale = np.nanmax(np.abs(ref[np.isfinite(ref)])) if np.isfinite(ref).any() else 1.0
            ok = np.isfinite(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-9, atol=1e-12 * scale, err_msg=f"{v} {name}")
            assert np.array_equal(np.isnan(got), np.isnan(ref)) or name == "R2", (v, name)
        at += ln
    df_var, df_idx = gm.metrics_tables(p, y, x)
    ref_var = d.metrics_var_scoring["m"].astype(np.float64)
    ref_idx = d.metrics_idx_scoring["m"].astype(np.float64)
    for name in ("MAE", "RMSE", "bias"):
        np.testing.assert_allclose(df_var[name].values, ref_var[name].values, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(df_idx[name].values, ref_idx[name].values, rtol=1e-9, atol=1e-14)
    fin = np.isfinite(ref_idx["R2"].values)
    np.testing.assert_allclose(df_idx["R2"].values[fin], ref_idx["R2"].values[fin], rtol=1e-8, atol=1e-10)
