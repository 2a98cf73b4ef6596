"""CPU-side checks of the C-ABI: the library builds for gfx950, loads, exports every symbol that
include/climsim_hip.h declares, and refuses to run (loudly) without a GPU."""
import ctypes as C
import os
import re

import pytest

from climsim_amd import _lib, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def test_header_symbols_all_bound_and_exported(lib):
    header = open(os.path.join(REPO, "include", "climsim_hip.h")).read()
    declared = set(re.findall(r"\b(cs_[a-z_0-9]+)\s*\(", header))
    declared -= {"cs_status", "cs_act", "cs_opt"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_cfg_struct_layout_matches_header():
    # 26 int32/float fields (2 + 16 hidden + 8) then 4 doubles, naturally aligned, no padding
    assert C.sizeof(_lib.CsMlpCfg) == 4 * 26 + 8 * 4
    assert C.sizeof(_lib.CsCnnCfg) == 4 * 14 + 8 * 4 + 8          # struct cs_cnn_cfg
    assert _lib.CsMlpCfg.beta1.offset == 104 and _lib.CsMlpCfg.flags.offset == 100


def test_version_and_error_string(lib):
    assert b"gfx950" in lib.cs_version()
    assert isinstance(lib.cs_last_error(), bytes)


def test_invalid_config_is_rejected_before_touching_the_device(lib):
    h = C.c_void_p()
    cfg = _lib.CsMlpCfg()
    cfg.n_in, cfg.n_hidden, cfg.n_out_lin, cfg.n_out_relu, cfg.max_batch = 124, 1, 120, 8, 128
    cfg.hidden[0] = 100                       # not a multiple of 128
    assert lib.cs_mlp_create(C.byref(h), C.byref(cfg)) == -1
    assert b"multiple of 128" in lib.cs_last_error()
    cfg.hidden[0] = 128
    cfg.n_out_relu = 9
    assert lib.cs_mlp_create(C.byref(h), C.byref(cfg)) == -1
    assert lib.cs_mlp_create(None, None) == -1
    assert lib.cs_mlp_num_params(None) == 0
    lib.cs_mlp_destroy(None)                  # no-op


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from climsim_amd.mlp import MLPEmulator
    with pytest.raises(_lib.EngineError):
        MLPEmulator(units=(128, 128))
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.CsMlpCfg()
    cfg.n_in, cfg.n_hidden, cfg.n_out_lin, cfg.n_out_relu, cfg.max_batch = 124, 1, 120, 8, 128
    cfg.hidden[0] = 128
    assert lib.cs_mlp_create(C.byref(h), C.byref(cfg)) == -2      # CS_ERR_HIP, not a CPU path
    assert not h.value


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(REPO, "climsim_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
