"""climsim_amd.data_utils against golden vectors produced by the reference's own data_utils
(tests/golden/make_golden.py).  float64 paths must agree to 1e-12
relative; they are the same numpy operations in the same order."""
import os

import numpy as np
import pytest

from climsim_amd.data_utils import data_utils
from golden_inputs import make_metric_inputs, make_crps_inputs, make_cnn_inputs, subsample

RTOL = 1e-12


@pytest.fixture(scope="module")
def gold(golden_dir):
    z = np.load(os.path.join(golden_dir, "data_utils_golden.npz"))
    return {k: z[k] for k in z.files}


def build(assets, **kw):
    import copy
    grid, *sets = assets
    return data_utils(copy.copy(grid), *sets, ml_backend="pytorch", **kw)


def check_big(gold, key, arr):
    arr = np.asarray(arr, dtype=np.float64)
    assert tuple(gold[key + "@shape"]) == arr.shape
    np.testing.assert_allclose(subsample(arr), gold[key + "@sub"], rtol=RTOL, atol=0)
    mom = np.array([arr.sum(), (arr * arr).sum(), np.abs(arr).max()])
    np.testing.assert_allclose(mom, gold[key + "@mom"], rtol=1e-10)


def test_ctor_attributes(lowres_assets, gold):
    d = build(lowres_assets)
    assert d.num_levels == 60 and d.num_latlon == 384
    np.testing.assert_allclose(d.area_wgt, gold["attr_area_wgt"], rtol=RTOL)
    np.testing.assert_array_equal(d.lats, gold["attr_lats"])
    np.testing.assert_array_equal(d.lons, gold["attr_lons"])
    assert d.ps_index is None and d.input_feature_len is None


@pytest.mark.parametrize("setter,n_in,n_out,ps", [("set_to_v1_vars", 124, 128, 120), ("set_to_v2_vars", 557, 368, 360),
                                                  ("set_to_v2_rh_vars", 557, 368, 360), ("set_to_v4_vars", 1525, 368, 1500),
                                                  ("set_to_v5_vars", 1405, 308, 1380)])
def test_variable_sets(lowres_assets, setter, n_in, n_out, ps):
    d = build(lowres_assets)
    getattr(d, setter)()
    assert (d.input_feature_len, d.target_feature_len, d.ps_index) == (n_in, n_out, ps)
    # the hard-coded lengths of the reference equal the sum of per-variable lengths
    assert sum(d.var_lens[v] for v in d.input_vars) == n_in
    assert sum(d.var_lens[v] for v in d.target_vars) == n_out


def test_g1_save_norm(lowres_assets, gold, tmp_path):
    d = build(lowres_assets)
    d.set_to_v1_vars()
    sub, div, scale = d.save_norm(str(tmp_path), write=True)
    np.testing.assert_array_equal(sub, gold["g1_v1_input_sub"])
    np.testing.assert_array_equal(div, gold["g1_v1_input_div"])
    np.testing.assert_array_equal(scale, gold["g1_v1_out_scale"])
    assert sub.shape == (124,) and scale.shape == (128,) and np.all(div != 0)
    # text format: one line, %.6e, comma separated
    line = open(tmp_path / "inp_sub.txt").read().strip()
    assert line.count(",") == 123 and line.split(",")[0] == "%.6e" % sub[0]
    d.set_to_v2_vars()
    sub, div, scale = d.save_norm()
    np.testing.assert_array_equal(sub, gold["g1_v2_input_sub"])
    np.testing.assert_array_equal(div, gold["g1_v2_input_div"])
    np.testing.assert_array_equal(scale, gold["g1_v2_out_scale"])
    assert (div == 0).sum() == 66  # CH4/N2O max==min levels: the inf/nan->0 rule is load-bearing for v2


def scoring_setup(assets):
    d = build(assets)
    d.set_to_v1_vars()
    x, y, p = make_metric_inputs(4)
    d.input_scoring, d.target_scoring = x, y
    d.set_pressure_grid("scoring")
    d.model_names = ["MLP"]
    d.preds_scoring = {"MLP": p}
    d.reweight_target("scoring")
    d.reweight_preds("scoring")
    return d, x, y, p


def test_g2_pressure_grid_and_weighting(lowres_assets, gold):
    d, x, y, p = scoring_setup(lowres_assets)
    check_big(gold, "g2_dp", d.dp_scoring)
    for v in d.target_vars:
        check_big(gold, f"g2_target_weighted/{v}", d.target_weighted_scoring[v])
        check_big(gold, f"g2_preds_weighted/{v}", d.preds_weighted_scoring["MLP"][v])


def test_g3_metrics_and_tables(lowres_assets, gold):
    d, *_ = scoring_setup(lowres_assets)
    d.metrics_names = ["MAE", "RMSE", "R2", "bias"]
    with np.errstate(all="ignore"):
        d.create_metrics_df("scoring")
        for v in d.target_vars:
            for m in d.metrics_names:
                fn = d.metrics_dict[m]
                a, b = d.preds_weighted_scoring["MLP"][v], d.target_weighted_scoring[v]
                np.testing.assert_allclose(fn(a, b), gold[f"g3_{m}/{v}"], rtol=RTOL, equal_nan=True)
                np.testing.assert_allclose(fn(a, b, avg_grid=False), gold[f"g3_{m}_grid/{v}"], rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(d.metrics_var_scoring["MLP"].to_numpy(dtype=np.float64), gold["g3_df_var"],
                               rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(d.metrics_idx_scoring["MLP"].to_numpy(dtype=np.float64), gold["g3_df_idx"],
                               rtol=RTOL, equal_nan=True)
    assert list(d.metrics_var_scoring["MLP"].index) == d.target_vars
    assert d.metrics_idx_scoring["MLP"].shape == (128, 4)


def test_g6_just_weights(lowres_assets, gold):
    d, x, y, p = scoring_setup(lowres_assets)
    check_big(gold, "g6_weights", d.output_weighting(y, "scoring", just_weights=True))


def test_g4_crps(lowres_assets, gold):
    d = build(lowres_assets)
    sp3, t3, sp2, t2 = make_crps_inputs()
    np.testing.assert_allclose(d.calc_CRPS(sp3, t3), gold["g4_crps_3d"], rtol=RTOL)
    np.testing.assert_allclose(d.calc_CRPS(sp3, t3, avg_grid=False), gold["g4_crps_3d_grid"], rtol=RTOL)
    np.testing.assert_allclose(d.calc_CRPS(sp2, t2), gold["g4_crps_2d"], rtol=RTOL)


def test_g5_cnn_reshapes(gold):
    xi, yi = make_cnn_inputs()
    xc = data_utils.reshape_input_for_cnn(xi)
    yc = data_utils.reshape_target_for_cnn(yi)
    np.testing.assert_array_equal(xc, gold["g5_input_cnn"])
    np.testing.assert_array_equal(yc, gold["g5_target_cnn"])
    assert xc.shape == (96, 60, 6) and yc.shape == (96, 60, 10)
    np.testing.assert_array_equal(data_utils.reshape_target_from_cnn(yc), gold["g5_target_back"])
    yc2 = yc + np.linspace(0, 1, 60, dtype=yc.dtype)[None, :, None]
    np.testing.assert_array_equal(data_utils.reshape_target_from_cnn(yc2), gold["g5_target_back_mean"])
    # round trip: profiles exact, scalars up to the float32 level-mean of 60 identical values
    np.testing.assert_allclose(data_utils.reshape_target_from_cnn(yc), yi, rtol=5e-6)


def test_split_errors(lowres_assets):
    d = build(lowres_assets)
    with pytest.raises(AssertionError):
        d.set_regexps("holdout", ["*"])
    with pytest.raises(AssertionError):
        d.get_filelist("train")
    d.data_path = "/nonexistent/"
    d.set_regexps("val", ["x*.nc"])
    with pytest.raises(AssertionError):
        d.set_filelist("val")          # stride not set
    d.set_stride_sample("val", 7)
    d.set_filelist("val")
    assert d.get_filelist("val") == []
