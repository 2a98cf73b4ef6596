"""ctypes binding of include/climsim_hip.h.  There is NO CPU fallback: if the shared library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CLIMSIM_HIP_LIB: another build of the same library (tools/sanitize_host.sh points it at the host-only ASan/UBSan build)
LIB_PATH = os.environ.get("CLIMSIM_HIP_LIB") or os.path.join(HERE, "libclimsim_hip.so")
CS_MAX_HIDDEN = 16
CS_FLAG_NO_TR_READ = 1
CS_FLAG_NO_CHAIN = 2
CS_FLAG_CHAIN_BM64 = 4
CS_FLAG_CHAIN_BM128 = 8
CS_FLAG_CHAIN_BM32 = 16
CS_FLAG_CHAIN_BWD32_ON_FWD64 = 32
CS_FLAG_GEMM_V1 = 64
CS_FLAG_NO_CHAIN_FB = 256
CS_FLAG_COOP = 512

ACT = {"relu": 0, "elu": 1, "leakyrelu": 2}
OPT = {"Adam": 0, "RAdam": 1, "RMSprop": 2, "SGD": 3, "AdamTorch": 4}
LOSS = {"mse": 0, "mae": 1, "huber": 2}
FLAG_DIRECT_HEAD = 128


class CsMlpCfg(C.Structure):
    _fields_ = [("n_in", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32 * CS_MAX_HIDDEN),
                ("n_out_lin", C.c_int32), ("n_out_relu", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
                ("optimizer", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("rho", C.c_double)]


CS_K_COUNT = 9
KERNEL_KINDS = ["prepare_input", "gemm_fwd", "gemm_dgrad", "wgrad", "optimizer", "memset", "chain_fwd", "chain_bwd", "chain_fb"]


class CsKernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * CS_K_COUNT), ("launches", C.c_int32 * CS_K_COUNT)]


class CsCnnCfg(C.Structure):
    _fields_ = ([(k, C.c_int32) for k in ("depth", "channels", "kernel", "seq", "c_in", "c_out", "n_lin", "max_batch",
                                          "device", "flags", "train", "optimizer", "loss", "reserved")]
                + [(k, C.c_double) for k in ("dropout", "beta1", "beta2", "eps")] + [("seed", C.c_uint64)])


class EngineError(RuntimeError):
    pass


_P, _I64, _I32, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_float
# name -> (restype, argtypes); must list every symbol include/climsim_hip.h declares
SIGNATURES = {
    "cs_mlp_create": (C.c_int, [C.POINTER(_P), C.POINTER(CsMlpCfg)]),
    "cs_mlp_destroy": (None, [_P]),
    "cs_mlp_num_params": (_I64, [_P]),
    "cs_mlp_device_bytes": (_I64, [_P]),
    "cs_mlp_check": (C.c_int, [_P, _P]),
    "cs_mlp_set_train_accuracy": (C.c_int, [_P, _P]),
    "cs_mlp_coop_timeouts": (_I64, [_P]),
    "cs_dp_unique_id": (C.c_int, [C.c_char_p, _P]),
    "cs_dp_init": (C.c_int, [C.POINTER(_P), C.c_char_p, _P, C.c_int, C.c_int, C.c_int]),
    "cs_dp_allreduce": (C.c_int, [_P, _P, C.c_int64, _P]),
    "cs_dp_allreduce_bf16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "cs_dp_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cs_dp_destroy": (None, [_P]),
    "cs_dp_ipc_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, _I64]),
    "cs_dp_ipc_export": (C.c_int, [_P, _P]),
    "cs_dp_ipc_connect": (C.c_int, [_P, _P]),
    "cs_dp_ipc_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "cs_dp_ipc_allreduce": (C.c_int, [_P, _I64, _P]),
    "cs_dp_ipc_timeouts": (_I64, [_P]),
    "cs_dp_ipc_set_timeout_ms": (C.c_int, [_P, C.c_double]),
    "cs_dp_ipc_destroy": (None, [_P]),
    "cs_mlp_set_norm": (C.c_int, [_P, _P, _P]),
    "cs_mlp_set_head_options": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "cs_mlp_set_dropout": (C.c_int, [_P, C.c_double, C.c_uint64]),
    "cs_mlp_set_weights": (C.c_int, [_P, _P, _I64, _P]),
    "cs_mlp_get_weights": (C.c_int, [_P, _P, _I64, _P]),
    "cs_mlp_get_opt_state": (C.c_int, [_P, _P, _P, _I64, C.POINTER(_I64), _P]),
    "cs_mlp_set_opt_state": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "cs_mlp_forward": (C.c_int, [_P, _P, _P, _I64, C.c_int, _P, _P, _P, C.c_int, _P]),
    "cs_mlp_loss_grads": (C.c_int, [_P, _P, _P, _P, _I64, C.c_int, _P, C.c_int, _P]),
    "cs_mlp_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "cs_mlp_set_grad_buffer": (C.c_int, [_P, _P]),
    "cs_mlp_get_grads": (C.c_int, [_P, _P, _I64, _P]),
    "cs_mlp_apply": (C.c_int, [_P, _F, _F, _P]),
    "cs_mlp_train_step": (C.c_int, [_P, _P, _P, _P, _I64, C.c_int, _F, _P, _P]),
    "cs_profile_begin": (C.c_int, [_P]),
    "cs_profile_end": (C.c_int, [C.POINTER(CsKernelTimes)]),
    "cs_mlp_profile_step": (C.c_int, [_P, _P, _P, _P, _I64, C.c_int, _F, _P, _P, C.POINTER(CsKernelTimes)]),
    "cs_mlp_debug_stamps": (C.c_int, [_P, _P, _I64]),
    "cs_mlp_debug_stamps_wgrad": (C.c_int, [_P, _P, _I64, _P]),
    "cs_cnn_debug_stamps": (C.c_int, [_P, _P, _I64, _P]),
    "cs_normalise_rows": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P]),
    "cs_permutation": (C.c_int, [_I64, C.c_uint64, _P, _P]),
    "cs_metrics_columns": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "cs_metrics_columns_x": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _I32, _I32, C.c_double, C.c_double, _P, _P, _P, _P, _P]),
    "cs_mlp_kernel_family": (C.c_int, [_P]),
    "cs_mlp_forward_limit": (_I64, [_P]),
    "cs_mlp_group_create": (C.c_int, [C.POINTER(_P), C.POINTER(_P), _I32]),
    "cs_mlp_group_destroy": (None, [_P]),
    "cs_mlp_group_size": (_I32, [_P]),
    "cs_mlp_group_train_step": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.c_int, C.POINTER(_F), _P, _P]),
    "cs_mlp_group_forward": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.c_int, C.POINTER(_P), C.POINTER(_P), _P, C.c_int, _P]),
    "cs_mlp_group_profile_step": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.c_int, C.POINTER(_F), _P, _P,
                                            C.POINTER(CsKernelTimes)]),
    "cs_categorical_accuracy": (C.c_int, [_P, _P, _I64, _I32, _P, C.c_int, _P]),
    "cs_loader_stack": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "cs_cnn_create": (C.c_int, [C.POINTER(_P), C.POINTER(CsCnnCfg)]),
    "cs_cnn_destroy": (None, [_P]),
    "cs_cnn_num_params": (_I64, [_P]),
    "cs_cnn_set_weights": (C.c_int, [_P, _P, _I64, _P]),
    "cs_cnn_get_weights": (C.c_int, [_P, _P, _I64, _P]),
    "cs_cnn_get_opt_state": (C.c_int, [_P, _P, _P, _I64, C.POINTER(_I64), _P]),
    "cs_cnn_set_opt_state": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "cs_cnn_forward": (C.c_int, [_P, _P, C.c_int, _I64, _P, _P, _P]),
    "cs_cnn_evaluate": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _I64, _P, C.c_int, _P]),
    "cs_cnn_loss_grads": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _I64, _P, _P]),
    "cs_cnn_set_seed": (C.c_int, [_P, C.c_uint64]),
    "cs_cnn_set_metrics_buffer": (C.c_int, [_P, _P]),
    "cs_cnn_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "cs_cnn_set_grad_buffer": (C.c_int, [_P, _P, _I64]),
    "cs_cnn_apply": (C.c_int, [_P, _F, _F, _P]),
    "cs_cnn_train_step": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _I64, _F, _P, _P]),
    "cs_last_error": (C.c_char_p, []),
    "cs_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libclimsim_hip.so (after torch, so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"{LIB_PATH} not found - build it with `python -m climsim_amd.build` "
                          "(there is no CPU fallback for the engine)")
    try:
        import torch  # noqa: F401  (its bundled libamdhip64 must be the one already mapped)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class profile_session:
    """`with profile_session(stream_ptr) as p: ...steps...` then `p.times` = {kind: (milliseconds, launches)} summed over every
    engine kernel launched in the block (cs_profile_begin / cs_profile_end; no synchronisation between the steps)."""

    def __init__(self, stream):
        self.stream, self.times = stream, None

    def __enter__(self):
        check(load().cs_profile_begin(self.stream))
        return self

    def __exit__(self, *exc):
        kt = CsKernelTimes()
        rc = load().cs_profile_end(C.byref(kt))
        if exc[0] is None:
            check(rc)
        self.times = {k: (float(kt.ms[i]), int(kt.launches[i])) for i, k in enumerate(KERNEL_KINDS)}
        return False


def check(rc: int):
    if rc != 0:
        raise EngineError(f"climsim_hip error {rc}: {load().cs_last_error().decode()}")
