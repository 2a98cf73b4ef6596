"""tools/accept_real.py (SURVEY 8(d) "MAE acceptance" on the real low-res splits) - the parts that need neither the data set nor a
GPU: the on-disk contract of the split files, the comparison against the published table, the verdict and the printed table."""
import importlib.util
import os

import numpy as np
import pandas as pd
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("accept_real", os.path.join(REPO, "tools", "accept_real.py"))
A = importlib.util.module_from_spec(spec)
spec.loader.exec_module(A)


def table(mae, r2, rmse):
    return pd.DataFrame({"MAE": mae, "R2": r2, "RMSE": rmse, "bias": np.zeros(10)}, index=list(A.VARS))


def test_published_tables_are_the_reference_figures():
    # website/evaluating.md:17-54 and the 16-digit MAE of tests/unit_tests.ipynb agree to the digits the web page prints
    assert len(A.VARS) == 10 and all(len(A.PUBLISHED[m][k]) == 10 for m in ("mlp", "cnn") for k in ("MAE", "R2", "RMSE"))
    for web, exact in zip(A.PUBLISHED["mlp"]["MAE"], A.PUBLISHED_MLP_MAE_EXACT):
        assert abs(web - exact) <= 0.6 * 10 ** (-(len(str(web).split(".")[1])))
    assert A.SPLIT_ROWS == {"train": 10091520, "val": 1441920, "scoring": 1681920} and all(n % 384 == 0 for n in A.SPLIT_ROWS.values())


def test_comparison_rows_and_verdict():
    pub = A.PUBLISHED["mlp"]
    ours = table(np.array(pub["MAE"]) * 1.004, [np.nan if p is None else p for p in pub["R2"]], np.array(pub["RMSE"]) * 0.99)
    rows = A.comparison_rows(ours, "mlp")
    assert len(rows) == 30
    r = next(x for x in rows if x["variable"] == "cam_out_PRECC" and x["metric"] == "MAE")
    assert r["published"] == 34.33 and r["rel_diff"] == pytest.approx(0.004, rel=1e-6)
    # metrics the published table does not report ("--") are printed without a difference and never gate
    q = next(x for x in rows if x["variable"] == "ptend_q0001" and x["metric"] == "R2")
    assert q["published"] is None and q["rel_diff"] is None and q["ours"] is None
    v = A.verdict(rows, 0.01)
    assert v["passed"] and v["n_checked"] == 10 and v["worst_rel_diff"] == pytest.approx(0.004, rel=1e-6)
    worse = table(np.array(pub["MAE"]) * np.array([1, 1, 1, 1.03, 1, 1, 1, 1, 1, 1]), np.zeros(10), np.array(pub["RMSE"]))
    v = A.verdict(A.comparison_rows(worse, "mlp"), 0.01)
    assert not v["passed"] and v["worst_variable"] == "cam_out_FLWDS" and v["worst_rel_diff"] == pytest.approx(0.03, rel=1e-6)
    text = A.format_table(rows, "mlp")
    assert text.count("\n") == 11 and "cam_out_SOLLD" in text and "34.33" in text and "column MLP" in text
    assert A.comparison_rows(table(pub["MAE"], np.zeros(10), pub["RMSE"]), "cnn")[0]["published"] == 2.585


def test_split_files_are_checked_against_the_on_disk_contract(tmp_path):
    rng = np.random.default_rng(0)
    for split, n in (("train", 768), ("val", 384)):
        np.save(tmp_path / f"{split}_input.npy", rng.normal(size=(n, 124)).astype(np.float32))
        np.save(tmp_path / f"{split}_target.npy", rng.normal(size=(n, 128)).astype(np.float32))
    x, y = A.load_split(str(tmp_path), "train")
    assert x.shape == (768, 124) and y.shape == (768, 128) and x.dtype == np.float32
    assert A.load_split(str(tmp_path), "train", limit_rows=500)[0].shape == (384, 124)       # whole time steps only
    with pytest.raises(FileNotFoundError, match="scoring_input.npy"):
        A.load_split(str(tmp_path), "scoring")
    np.save(tmp_path / "scoring_input.npy", rng.normal(size=(384, 124)))                      # float64: the loader's output is float32
    np.save(tmp_path / "scoring_target.npy", rng.normal(size=(384, 128)).astype(np.float32))
    with pytest.raises(ValueError, match="float32"):
        A.load_split(str(tmp_path), "scoring")
    np.save(tmp_path / "scoring_input.npy", rng.normal(size=(385, 124)).astype(np.float32))
    with pytest.raises(ValueError, match="rows"):
        A.load_split(str(tmp_path), "scoring")
    np.save(tmp_path / "val_target.npy", rng.normal(size=(384, 120)).astype(np.float32))
    with pytest.raises(ValueError, match="expected"):
        A.load_split(str(tmp_path), "val")
