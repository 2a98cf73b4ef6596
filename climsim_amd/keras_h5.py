"""Keras `.h5` checkpoints of the baseline MLP, read and written without TensorFlow / h5py.

The reference checkpoints with `keras.callbacks.ModelCheckpoint(filepath=...h5, save_weights_only=False)`
(baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:253-261), continues training from
`keras.models.load_model(... .h5)` (:134-136) and publishes `baseline_models/MLP/model/backup_phase-7_retrained_models_
step2_lot-147_trial_0027.best.h5`.  Such a file is HDF5 written by h5py in Keras' "legacy H5" layout:

    /  (attrs keras_version, backend, model_config, training_config)
    /model_weights             attrs layer_names = [b'input', b'dense', b'leaky_re_lu', ...]   (creation order)
    /model_weights/<layer>     attrs weight_names = [b'<layer>/kernel:0', b'<layer>/bias:0'] (empty for weight-less layers)
    /model_weights/<layer>/<layer>/kernel:0 (in, out) float32,  bias:0 (out,) float32
    /optimizer_weights/...

(`model.save_weights(x.h5)` writes the content of /model_weights at the root.)  `load_keras_h5` returns the weight list in
the order Keras' `model.get_weights()` has - the order `MLPEmulator.set_weights` takes - by following `layer_names` and
`weight_names`; `save_keras_h5` writes that layout for the layer sequence of `step2_retrain.build_model` (:95-128): the
input layer 'input' and the model name 'retrained_model' as the reference names them, Keras' automatic names for every other
layer, so `model.load_weights(path)` of a reference-built model finds every tensor where it looks
(keras/saving/legacy/hdf5_format.py `load_weights_from_hdf5_group`: layers with weights are matched in order, tensors by
`weight_names`).  Pinned by files the HDF5 C library wrote in this layout (tests/golden/hdf5/make_keras_h5_fixture.c) and by
reading our own files back with `h5dump`; NOT verified against a running Keras (TensorFlow is not installable here).
"""
from __future__ import annotations

import json
from typing import List, Optional, Sequence

import numpy as np

from .hdf5 import ATTRS, Hdf5File, write_hdf5_tree

_ACT_LAYER = {"relu": "re_lu", "elu": "elu", "leakyrelu": "leaky_re_lu"}
_ACT_CLASS = {"relu": "ReLU", "elu": "ELU", "leakyrelu": "LeakyReLU"}


# step2_retrain.build_model names exactly two things itself: `keras.layers.Input(..., name='input')` (:96) and
# `keras.Model(..., name='retrained_model')` (:128); every other layer gets Keras' automatic name.
INPUT_LAYER = "input"
MODEL_NAME = "retrained_model"


def _numbered(base: str, i: int) -> str:
    return base if i == 0 else f"{base}_{i}"


def keras_layer_sequence(n_hidden: int, activation: str):
    """[(layer name, has weights)] in creation order for build_model (step2_retrain.py:95-126): Input, n_hidden x (Dense,
    activation), Dense(output_length), activation, Dense(n_lin), Dense(n_relu, relu), Concatenate."""
    act = _ACT_LAYER[activation]
    seq = [(INPUT_LAYER, False)]
    for i in range(n_hidden + 1):
        seq += [(_numbered("dense", i), True), (_numbered(act, i), False)]
    seq += [(_numbered("dense", n_hidden + 1), True), (_numbered("dense", n_hidden + 2), True), ("concatenate", False)]
    return seq


def model_config_json(weights: Sequence[np.ndarray], activation: str, alpha: float = 0.15) -> str:
    """Functional-model config of build_model for these weight shapes (what `model.to_json()` holds, abridged to the fields
    Keras' deserialiser needs).  Written for completeness of the full-model file; untested against Keras."""
    n_hidden = len(weights) // 2 - 3
    seq = keras_layer_sequence(n_hidden, activation)
    n_in = int(weights[0].shape[0])
    layers = [{"class_name": "InputLayer", "name": INPUT_LAYER, "inbound_nodes": [],
               "config": {"batch_input_shape": [None, n_in], "dtype": "float32", "sparse": False, "ragged": False, "name": INPUT_LAYER}}]
    prev, wi = INPUT_LAYER, 0
    dense_names = [n for n, w in seq if w]
    for name, has_w in seq[1:-3]:
        if has_w:
            units = int(weights[2 * wi].shape[1])
            wi += 1
            cfg = {"name": name, "trainable": True, "dtype": "float32", "units": units, "activation": "linear", "use_bias": True}
            layers.append({"class_name": "Dense", "name": name, "config": cfg, "inbound_nodes": [[[prev, 0, 0, {}]]]})
        else:
            cfg = {"name": name, "trainable": True, "dtype": "float32"}
            if activation == "leakyrelu":
                cfg["alpha"] = float(alpha)
            layers.append({"class_name": _ACT_CLASS[activation], "name": name, "config": cfg, "inbound_nodes": [[[prev, 0, 0, {}]]]})
        prev = name
    lin, rel = dense_names[-2], dense_names[-1]
    for name, act_name, w in ((lin, "linear", weights[-4]), (rel, "relu", weights[-2])):
        layers.append({"class_name": "Dense", "name": name, "inbound_nodes": [[[prev, 0, 0, {}]]],
                       "config": {"name": name, "trainable": True, "dtype": "float32", "units": int(w.shape[1]), "activation": act_name, "use_bias": True}})
    layers.append({"class_name": "Concatenate", "name": "concatenate", "config": {"name": "concatenate", "trainable": True, "dtype": "float32", "axis": -1},
                   "inbound_nodes": [[[lin, 0, 0, {}], [rel, 0, 0, {}]]]})
    return json.dumps({"class_name": "Functional", "config": {"name": MODEL_NAME, "layers": layers, "input_layers": [[INPUT_LAYER, 0, 0]],
                                                              "output_layers": [["concatenate", 0, 0]]}})


def save_keras_h5(path: str, weights: Sequence[np.ndarray], activation: str = "leakyrelu", alpha: float = 0.15,
                  full_model: bool = True, optimizer_state: Optional[dict] = None) -> None:
    """Write `weights` (Keras order [W0, b0, ..., W_lin, b_lin, W_relu, b_relu], kernels (in, out)) in Keras' legacy-H5 layout.
    `full_model`: the layout of `model.save(path)` (root attrs + /model_weights), else that of `model.save_weights(path)`.
    `optimizer_state` = {'m': [...], 'v': [...], 'iterations': int} goes to /optimizer_weights/climsim_amd (this package's own
    slot layout; Keras ignores it when it loads weights).
    Intended compatibility: `model.load_weights(path)` on a model built by the reference's `build_model`.  NOT supported:
    resuming through `keras.models.load_model(path)` (the reference's sw_continue branch, step2_retrain.py:134-136) - the
    `model_config` JSON is unverified against Keras and the optimiser slots are not in Keras' `optimizer_weights` naming
    ('Adam/dense/kernel/m:0', ..., 'iter'), so a Keras resume would start Adam from zero.  Resume with this package instead
    (`MLPEmulator.load_weights` restores weights, both moments and the iteration count)."""
    if len(weights) < 8 or len(weights) % 2:
        raise ValueError("expected the Keras weight list of the baseline MLP: >= 1 hidden layer, the 128-wide layer and two heads")
    n_hidden = len(weights) // 2 - 3
    seq = keras_layer_sequence(n_hidden, activation)
    grp = {ATTRS: {"layer_names": np.asarray([n.encode() for n, _ in seq]), "backend": "tensorflow", "keras_version": "2.11.0"}}
    wi = 0
    for name, has_w in seq:
        if has_w:
            k, b = np.asarray(weights[2 * wi], np.float32), np.asarray(weights[2 * wi + 1], np.float32)
            wi += 1
            grp[name] = {ATTRS: {"weight_names": np.asarray([f"{name}/kernel:0".encode(), f"{name}/bias:0".encode()])},
                         name: {"kernel:0": k, "bias:0": b}}
        else:
            grp[name] = {ATTRS: {"weight_names": np.asarray([], dtype=np.float64)}}        # what np.asarray([]) gives Keras
    if not full_model:
        tree = grp
    else:
        tree = {ATTRS: {"keras_version": "2.11.0", "backend": "tensorflow", "model_config": model_config_json(weights, activation, alpha)},
                "model_weights": grp}
    if optimizer_state is not None:
        ow = {"iterations": np.asarray(int(optimizer_state["iterations"]), np.int64)}
        for i, (m, v) in enumerate(zip(optimizer_state["m"], optimizer_state["v"])):
            ow[f"m_{i}"] = np.asarray(m, np.float32)
            ow[f"v_{i}"] = np.asarray(v, np.float32)
        tree["optimizer_weights"] = {ATTRS: {"weight_names": np.asarray([], dtype=np.float64)}, "climsim_amd": ow}
    write_hdf5_tree(path, tree)


def load_keras_h5(path: str, with_optimizer: bool = False):
    """Weight list in `model.get_weights()` order from a Keras legacy-H5 file (full model or weights only).  With
    `with_optimizer`, also the optimiser slots this package wrote (None for a file from Keras)."""
    with Hdf5File(path) as f:
        root = "model_weights" if "model_weights" in f.groups() and "layer_names" not in f.attrs("") else ""
        top = f.attrs(root)
        if "layer_names" not in top:
            raise ValueError(f"{path}: no layer_names attribute - not a Keras .h5 checkpoint")
        weights: List[np.ndarray] = []
        for ln in np.atleast_1d(top["layer_names"]):
            lname = ln.decode() if isinstance(ln, bytes) else str(ln)
            g = f"{root}/{lname}".strip("/")
            names = np.atleast_1d(f.attrs(g).get("weight_names", np.asarray([])))
            for wn in names:
                if not isinstance(wn, (bytes, str, np.bytes_, np.str_)):
                    continue                                       # the empty float64 array of a weight-less layer
                wname = wn.decode() if isinstance(wn, bytes) else str(wn)
                weights.append(np.ascontiguousarray(f[f"{g}/{wname}"], dtype=np.float32))
        opt = None
        if with_optimizer and "optimizer_weights/climsim_amd" in f.groups():
            n = len(weights)
            opt = {"iterations": int(np.asarray(f["optimizer_weights/climsim_amd/iterations"]).ravel()[0]),
                   "m": [np.asarray(f[f"optimizer_weights/climsim_amd/m_{i}"], np.float32) for i in range(n)],
                   "v": [np.asarray(f[f"optimizer_weights/climsim_amd/v_{i}"], np.float32) for i in range(n)]}
    return (weights, opt) if with_optimizer else weights
