"""Argument validation of the C ABI (include/climsim_hip.h): every entry point must refuse null / out-of-range arguments with a
negative cs_status and a message - BEFORE any device work - and never crash.  Runs without a GPU; tools/sanitize_host.sh
runs this file (and test_cabi_cpu.py) against a host-only AddressSanitizer + UBSan build of the library (SURVEY section 5:
sanitizers on the CPU build only)."""
import ctypes as C

import pytest

from climsim_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def bad(lib, rc, needle=None):
    assert rc < 0, rc
    msg = lib.cs_last_error()
    assert isinstance(msg, bytes) and len(msg) > 0
    if needle:
        assert needle in msg, msg


def test_mlp_entries_refuse_null_handles_and_buffers(lib):
    z = C.c_void_p()
    f4 = (C.c_float * 4)()
    bad(lib, lib.cs_mlp_set_norm(None, f4, f4), b"null")
    bad(lib, lib.cs_mlp_set_head_options(None, 0, None, 0))
    bad(lib, lib.cs_mlp_set_dropout(None, 0.1, 0))
    bad(lib, lib.cs_mlp_set_weights(None, f4, 4, None))
    bad(lib, lib.cs_mlp_get_weights(None, f4, 4, None))
    bad(lib, lib.cs_mlp_get_opt_state(None, f4, f4, 4, None, None))
    bad(lib, lib.cs_mlp_set_opt_state(None, f4, f4, 4, 0, None))
    bad(lib, lib.cs_mlp_forward(None, f4, None, 4, 0, None, None, None, 0, None), b"null handle")
    bad(lib, lib.cs_mlp_loss_grads(None, f4, f4, None, 4, 0, f4, 0, None))
    bad(lib, lib.cs_mlp_train_step(None, f4, f4, None, 4, 0, 1e-3, f4, None))
    bad(lib, lib.cs_mlp_apply(None, 1e-3, 1.0, None))
    bad(lib, lib.cs_mlp_grad_buffer(None, C.byref(z), None))
    bad(lib, lib.cs_mlp_set_grad_buffer(None, None))
    bad(lib, lib.cs_mlp_get_grads(None, f4, 4, None))
    bad(lib, lib.cs_mlp_check(None, None), b"null handle")
    assert lib.cs_mlp_coop_timeouts(None) == 0
    assert lib.cs_mlp_device_bytes(None) == 0
    assert lib.cs_mlp_kernel_family(None) == -1
    assert lib.cs_mlp_forward_limit(None) == 0
    bad(lib, lib.cs_mlp_profile_step(None, f4, f4, None, 4, 0, 1e-3, f4, None, None))
    bad(lib, lib.cs_mlp_debug_stamps(None, None, 0))
    bad(lib, lib.cs_mlp_debug_stamps_wgrad(None, None, 0, None))
    bad(lib, lib.cs_cnn_debug_stamps(None, None, 0, None))
    bad(lib, lib.cs_profile_end(None))
    bad(lib, lib.cs_mlp_set_train_accuracy(None, None), b"null handle")
    i8 = (C.c_int64 * 8)()
    bad(lib, lib.cs_permutation(8, 1, None, None), b"null")
    bad(lib, lib.cs_permutation(0, 1, i8, None), b"outside")
    bad(lib, lib.cs_permutation((1 << 31) + 1, 1, i8, None), b"outside")
    bad(lib, lib.cs_dp_ipc_set_timeout_ms(None, 1.0))


def test_config_ranges(lib):
    h = C.c_void_p()
    cfg = _lib.CsMlpCfg()
    cfg.n_in, cfg.n_hidden, cfg.n_out_lin, cfg.n_out_relu, cfg.max_batch = 124, 1, 120, 8, 128
    cfg.hidden[0] = 128
    for field, value, needle in (("n_in", 0, b"n_in"), ("n_in", 5000, b"n_in"), ("n_hidden", 0, b"n_hidden"), ("n_hidden", 17, b"n_hidden"),
                                 ("n_out_lin", -4, b"heads"), ("n_out_lin", 1024, b"heads"), ("act", 3, b"activation"), ("optimizer", 5, b"optimizer"),
                                 ("max_batch", 0, b"max_batch")):
        keep = getattr(cfg, field)
        setattr(cfg, field, value)
        bad(lib, lib.cs_mlp_create(C.byref(h), C.byref(cfg)), needle)
        assert not h.value
        setattr(cfg, field, keep)


def test_group_loader_metrics_dp_and_cnn_entries(lib):
    g = C.c_void_p()
    P = C.c_void_p
    bad(lib, lib.cs_mlp_group_create(C.byref(g), None, 2), b"null")
    arr = (P * 2)(None, None)
    bad(lib, lib.cs_mlp_group_create(C.byref(g), arr, 0))
    bad(lib, lib.cs_mlp_group_create(C.byref(g), arr, 33))
    bad(lib, lib.cs_mlp_group_create(C.byref(g), arr, 2), b"member 0 is null")
    assert lib.cs_mlp_group_size(None) == 0
    lib.cs_mlp_group_destroy(None)
    n = (C.c_int64 * 2)(1, 1)
    lr = (C.c_float * 2)(0.0, 0.0)
    bad(lib, lib.cs_mlp_group_train_step(None, arr, arr, None, n, 0, lr, None, None))
    bad(lib, lib.cs_mlp_group_forward(None, arr, None, n, 0, None, None, None, 0, None))
    f8 = (C.c_double * 8)()
    bad(lib, lib.cs_loader_stack(None, None, 1, 1, 1, 1, f8, f8, 0, None, None, None, None, None), b"mli")
    bad(lib, lib.cs_loader_stack(f8, None, 1, 1, 1, 1, f8, f8, 0, None, None, None, None, None), b"no output")
    bad(lib, lib.cs_loader_stack(f8, None, 1, 70000, 1, 1, f8, f8, 0, None, None, f8, None, None), b"bad sizes")
    bad(lib, lib.cs_metrics_columns(None, None, 1, 1, 1, None, None, None, None, None, None))
    bad(lib, lib.cs_metrics_columns_x(None, None, 1, 1, 1, None, 124, 120, 1.0, 0.0, None, None, None, None, None))
    bad(lib, lib.cs_normalise_rows(None, None, 1, 1, None, None, None, None))
    bad(lib, lib.cs_categorical_accuracy(None, None, 1, 1, None, 0, None))
    bad(lib, lib.cs_dp_unique_id(b"/nonexistent/librccl.so", C.create_string_buffer(128)), b"cannot load RCCL")
    bad(lib, lib.cs_dp_unique_id(None, None))
    c = C.c_void_p()
    bad(lib, lib.cs_dp_init(C.byref(c), None, None, 1, 0, 0))
    bad(lib, lib.cs_dp_init(C.byref(c), None, C.create_string_buffer(128), 2, 5, 0), b"rank 5 of 2")
    bad(lib, lib.cs_dp_allreduce(None, None, 0, None))
    bad(lib, lib.cs_dp_allreduce_bf16(None, None, 0, None))
    i0, i1 = C.c_int(), C.c_int()
    bad(lib, lib.cs_dp_comm_info(None, C.byref(i0), C.byref(i1)))
    lib.cs_dp_destroy(None)
    ic = C.c_void_p()
    bad(lib, lib.cs_dp_ipc_create(None, 2, 0, 0, 1024))
    bad(lib, lib.cs_dp_ipc_create(C.byref(ic), 9, 0, 0, 1024), b"at most 8")
    bad(lib, lib.cs_dp_ipc_create(C.byref(ic), 2, 2, 0, 1024))
    bad(lib, lib.cs_dp_ipc_create(C.byref(ic), 2, 0, 0, 1023), b"multiple of 4")
    bad(lib, lib.cs_dp_ipc_export(None, None))
    bad(lib, lib.cs_dp_ipc_connect(None, None))
    bad(lib, lib.cs_dp_ipc_buffer(None, None, None))
    bad(lib, lib.cs_dp_ipc_allreduce(None, 4, None))
    assert lib.cs_dp_ipc_timeouts(None) == 0
    lib.cs_dp_ipc_destroy(None)
    h = C.c_void_p()
    bad(lib, lib.cs_cnn_create(C.byref(h), None))
    bad(lib, lib.cs_cnn_create(None, None))
    assert lib.cs_cnn_num_params(None) == 0
    lib.cs_cnn_destroy(None)
    f4 = (C.c_float * 4)()
    for rc in (lib.cs_cnn_set_weights(None, f4, 4, None), lib.cs_cnn_get_weights(None, f4, 4, None), lib.cs_cnn_forward(None, f4, 0, 1, None, None, None),
               lib.cs_cnn_evaluate(None, f4, 0, f4, 0, None, 1, f4, 0, None), lib.cs_cnn_loss_grads(None, f4, 0, f4, 0, None, 1, f4, None),
               lib.cs_cnn_set_seed(None, 1), lib.cs_cnn_set_metrics_buffer(None, None), lib.cs_cnn_apply(None, 1e-3, 1.0, None), lib.cs_cnn_train_step(None, f4, 0, f4, 0, None, 1, 1e-3, f4, None),
               lib.cs_cnn_set_grad_buffer(None, None, 0), lib.cs_cnn_grad_buffer(None, None, None),
               lib.cs_cnn_get_opt_state(None, f4, f4, 4, None, None), lib.cs_cnn_set_opt_state(None, f4, f4, 4, 0, None)):
        bad(lib, rc)
