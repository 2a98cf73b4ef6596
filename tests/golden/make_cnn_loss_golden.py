"""Golden vectors for the CNN's loss / metric FUNCTIONS, made by running the reference's own function bodies.

    python tests/golden/make_cnn_loss_golden.py          # needs /root/reference; writes tests/golden/cnn_loss_golden.npz

`baseline_models/CNN/training/hpo_train.py` cannot be imported here (tensorflow, keras, keras_tuner are absent).  Its three
loss / metric functions - `continuous_ranked_probability_score` (:83-111), `mse_adjusted` (:114-116), `mae_adjusted` (:119-121) - are
pure expressions over a handful of backend operations, so this script takes their SOURCE out of the reference file (ast: the three
FunctionDef nodes, nothing else of the module is executed) and runs it with `tf` / `K` bound to a namespace of numpy namesakes
(`reduce_mean`, `abs`, `subtract`, `add`, `multiply`, `expand_dims`, `constant`, `square`, `mean`): what is evaluated - which terms,
which axes, which weights 120/128 and 8/128, which slices - is the reference's own text; only the elementwise arithmetic is numpy's
(float64).  The fixture stores the outputs for seeded (B, 60, 10) tensors (tests/golden/cnn_loss_inputs in this file's `inputs()`);
tests/test_oracle.py holds oracle/cnn_oracle.py's restatements to them."""
import ast
import os
import types

import numpy as np

REF = "/root/reference/baseline_models/CNN/training/hpo_train.py"
NAMES = ("continuous_ranked_probability_score", "mse_adjusted", "mae_adjusted")


def inputs():
    rs = np.random.RandomState(20240611)
    out = {}
    for name, shape in (("a", (7, 60, 10)), ("b", (3, 60, 10))):
        yt = rs.standard_normal(shape) * 0.3
        yp = yt + rs.standard_normal(shape) * 0.1
        yp[..., 2:] = np.maximum(yp[..., 2:], 0)
        out[name] = (yt, yp)
    return out


def reference_functions():
    tree = ast.parse(open(REF).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in NAMES]
    assert sorted(n.name for n in body) == sorted(NAMES)
    tf = types.SimpleNamespace(
        reduce_mean=lambda x, axis=None: np.mean(x, axis=axis), abs=np.abs, subtract=np.subtract, add=np.add, multiply=np.multiply,
        expand_dims=np.expand_dims, constant=lambda v, dtype=None: np.asarray(v, dtype=dtype))
    K = types.SimpleNamespace(square=np.square, abs=np.abs, mean=lambda x: np.mean(x))
    ns = {"tf": tf, "K": K}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return {n: ns[n] for n in NAMES}


def main():
    fns = reference_functions()
    out = {}
    for key, (yt, yp) in inputs().items():
        for n, f in fns.items():
            out[f"{key}/{n}"] = np.float64(f(yt, yp))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cnn_loss_golden.npz")
    np.savez(path, **out)
    print(path, {k: float(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
