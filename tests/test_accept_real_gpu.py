"""tools/accept_real.py end to end on the GPU with made-up split files of the real layout: the published MLP configuration trains,
the scoring split is predicted and scored through the device metrics, the notebook's prediction file and the JSON report are
written.  (Random data: the verdict against the published table is, rightly, FAIL - exit code 1.)"""
import importlib.util
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_accept_real_runs_end_to_end_on_files_of_the_real_layout(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    spec = importlib.util.spec_from_file_location("accept_real", os.path.join(REPO, "tools", "accept_real.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    from oracle import mlp_oracle as O
    for split, n, seed in (("train", 32 * 384, 1), ("val", 2 * 384, 2), ("scoring", 2 * 384, 3)):
        x, y = O.synth_columns(n, seed=seed)
        np.save(tmp_path / f"{split}_input.npy", x.astype(np.float32))
        np.save(tmp_path / f"{split}_target.npy", y.astype(np.float32))
    out = tmp_path / "out"
    rc = A.main(["--data", str(tmp_path), "--model", "mlp", "--epochs", "2", "--out", str(out)])
    assert rc == 1                                                       # random columns do not reproduce the published MAE
    preds = np.load(out / "MLP_preds.npy")
    assert preds.shape == (768, 128) and preds.dtype == np.float32 and np.isfinite(preds).all() and (preds[:, 120:] >= 0).all()
    rep = json.load(open(out / "accept_mlp.json"))
    assert rep["model"] == "mlp" and len(rep["rows"]) == 30 and rep["verdict"]["passed"] is False and len(rep["history"]["val_loss"]) == 2
    assert rep["history"]["loss"][1] < rep["history"]["loss"][0]
    assert os.path.exists(out / "accept_mlp_best.npz") and open(out / "accept_mlp_log.csv").read().startswith("epoch,accuracy,loss")
