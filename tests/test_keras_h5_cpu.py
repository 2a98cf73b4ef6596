"""Keras `.h5` checkpoints (climsim_amd/keras_h5.py, hdf5.py attributes + tree writer) without TensorFlow / h5py.
Fixtures: files the HDF5 C library wrote in Keras' legacy-H5 layout (tests/golden/hdf5/make_keras_h5_fixture.c: h5py-style
variable-length string attributes, numpy 'S' arrays, nested layer groups, a null dataspace); our own files are read back
by the library's `h5dump` when it is installed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from climsim_amd.hdf5 import Hdf5File, write_hdf5_tree
from climsim_amd.keras_h5 import keras_layer_sequence, load_keras_h5, model_config_json, save_keras_h5

HERE = os.path.join(os.path.dirname(__file__), "golden", "hdf5")
H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def expected_fixture_weights():
    shapes = [(8, 128), (128, 8), (8, 4), (8, 4)]
    out = []
    for t, (k, n) in enumerate(shapes):
        out.append((0.001 * (t + 1) * np.arange(k * n) - 0.05 * (t + 1)).astype(np.float32).reshape(k, n))
        out.append((0.01 * (t + 1) * np.arange(n) - 0.02).astype(np.float32))
    return out


@pytest.mark.parametrize("name", ["keras_mlp_full_model.h5", "keras_mlp_weights_only.h5"])
def test_reads_libhdf5_written_keras_layout(name):
    path = os.path.join(HERE, name)
    ws = load_keras_h5(path)
    exp = expected_fixture_weights()
    assert len(ws) == len(exp)
    for a, b in zip(ws, exp):
        assert a.dtype == np.float32 and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    with Hdf5File(path) as f:
        root = "model_weights" if name.startswith("keras_mlp_full") else ""
        top = f.attrs(root)
        assert [n.decode() for n in top["layer_names"]] == ["input_1", "dense", "leaky_re_lu", "dense_1", "leaky_re_lu_1", "dense_2",
                                                            "dense_3", "concatenate"]
        assert top["backend"] == "tensorflow" and top["keras_version"] == "2.11.0"          # variable-length strings (global heap)
        assert f.attrs(f"{root}/concatenate".strip("/"))["weight_names"].size == 0          # null dataspace
        if root:
            assert "Functional" in f.attrs("")["model_config"]
            assert int(f["optimizer_weights/Adam/iter:0"]) == 59130


def test_round_trip_in_creation_order_with_many_layers(tmp_path):
    # 12 hidden layers (the search space's maximum): dense_10 sorts before dense_2 in the group's symbol table, layer_names does not
    rng = np.random.default_rng(0)
    dims = [124] + [128] * 12 + [128]
    ws = []
    for k, n in zip(dims[:-1], dims[1:]):
        ws += [rng.normal(size=(k, n)).astype(np.float32), rng.normal(size=n).astype(np.float32)]
    ws += [rng.normal(size=(128, 120)).astype(np.float32), rng.normal(size=120).astype(np.float32),
           rng.normal(size=(128, 8)).astype(np.float32), rng.normal(size=8).astype(np.float32)]
    seq = keras_layer_sequence(12, "elu")
    assert [n for n, w in seq if w][:3] == ["dense", "dense_1", "dense_2"] and seq[2][0] == "elu" and seq[4][0] == "elu_1"
    for full in (True, False):
        p = str(tmp_path / f"m{int(full)}.h5")
        save_keras_h5(p, ws, "elu", full_model=full, optimizer_state={"m": ws, "v": ws, "iterations": 7} if full else None)
        got, opt = load_keras_h5(p, with_optimizer=True)
        assert len(got) == len(ws) and all(np.array_equal(a, b) for a, b in zip(got, ws))
        assert (opt is not None and opt["iterations"] == 7) == full
    cfg = __import__("json").loads(model_config_json(ws, "elu"))
    names = [l["name"] for l in cfg["config"]["layers"]]
    assert names[0] == "input" and names[-1] == "concatenate" and names.count("dense_14") == 1       # step2_retrain.py:96 names the input layer
    assert cfg["config"]["name"] == "retrained_model"                                                 # step2_retrain.py:128
    with pytest.raises(ValueError):
        save_keras_h5(str(tmp_path / "bad.h5"), ws[:5])
    write_hdf5_tree(str(tmp_path / "plain.h5"), {"data": np.zeros(3, np.float32)})
    with pytest.raises(ValueError):
        load_keras_h5(str(tmp_path / "plain.h5"))


@pytest.mark.skipif(H5DUMP is None, reason="needs the HDF5 command line tools")
def test_written_checkpoint_is_read_by_libhdf5(tmp_path):
    ws = [np.full(s, i + 0.5, np.float32) for i, s in enumerate([(124, 128), (128,), (128, 128), (128,), (128, 120), (120,), (128, 8), (8,)])]
    p = str(tmp_path / "ck.h5")
    save_keras_h5(p, ws, "leakyrelu")
    out = subprocess.run([H5DUMP, "-A", p], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for token in ('ATTRIBUTE "layer_names"', '"leaky_re_lu_1"', 'GROUP "model_weights"', '"dense_2/kernel:0"', 'ATTRIBUTE "model_config"'):
        assert token in out.stdout, token
    one = subprocess.run([H5DUMP, "-d", "/model_weights/dense_3/dense_3/bias:0", p], capture_output=True, text=True)
    assert one.returncode == 0 and "H5T_IEEE_F32LE" in one.stdout and "( 8 )" in one.stdout and "7.5" in one.stdout
