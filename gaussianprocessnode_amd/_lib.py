"""ctypes binding of the C ABI declared in include/sgp_hip.h.

The product path has no CPU fallback: if the library is missing or no gfx950 device is visible the
calls raise.  (`oracle/` is test infrastructure and is never imported from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

SGP_FLAG_NO_GRAPH = 1
SGP_FLAG_KEEP_KUF = 2
SGP_FLAG_GRAPH = 4
SGP_FLAG_PERSISTENT_CHAIN = 8
SGP_S_YY, SGP_S_W, SGP_S_N, SGP_S_COUNT = 0, 1, 2, 8
(SGP_R_SUM_I1, SGP_R_SUM_I2, SGP_R_ENERGY, SGP_R_INFO_KUU, SGP_R_INFO_LAMBDA, SGP_R_INFO_PRIOR,
 SGP_R_LOGDET_KUU, SGP_R_LOGDET_LAMBDA, SGP_R_COUNT) = range(9)
SGP_T_SWEEP, SGP_T_GRAM, SGP_T_SYRK, SGP_T_FINISH1, SGP_T_FINISH2, SGP_T_GAP_LOCAL_FINISH, SGP_T_KUU, SGP_T_LOCAL = range(8)
SGP_T_COUNT = 8

EXPORTS = [
    "sgp_abi_version", "sgp_create", "sgp_destroy", "sgp_last_error", "sgp_set_inducing", "sgp_set_data",
    "sgp_set_kernel", "sgp_set_output_cov_sum", "sgp_set_prior", "sgp_set_noise", "sgp_sweep_local", "sgp_sweep_finish", "sgp_sweep",
    "sgp_stats_layout", "sgp_bind_stats", "sgp_get_posterior", "sgp_get_scalars", "sgp_get_stats",
    "sgp_get_kuu_chol", "sgp_get_wishart_invscale", "sgp_w_stats", "sgp_predict", "sgp_theta_objective", "sgp_carry_posterior", "sgp_set_posterior",
    "sgp_kernelmatrix", "sgp_potrf", "sgp_potri", "sgp_get_timestamps", "sgp_get_phase_totals", "sgp_time_kernel", "sgp_get_chain_trace", "sgp_set_allreduce", "sgp_use_rccl", "sgp_measure_sclk_mhz",
    "sgp_train_begin", "sgp_train_step", "sgp_train_end", "sgp_get_step_trace", "sgp_measure_clocks", "sgp_overlap_plan",
    "sgp_get_sweep_trace", "sgp_train_likelihood", "sgp_train_get_gamma", "sgp_wait",
]
SGP_TIME_GROUP0 = 100
SGP_TIME_QUADFORM = 120


class SGPError(RuntimeError):
    """Negative status from the C ABI (bad argument, HIP error, no device)."""


class PosDefException(ArithmeticError):
    """Positive status k: leading minor k is not positive definite (Julia's PosDefException(k))."""

    def __init__(self, k, msg=""):
        super().__init__(f"matrix is not positive definite; Cholesky failed at minor {k}. {msg}")
        self.info = k


class Config(C.Structure):
    _fields_ = [("n_max", C.c_int64), ("m", C.c_int32), ("d", C.c_int32), ("d_out", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32)]


# int hook(void* ctx, void* stats_dev, int64_t count, void* stream): the all-reduce step of sgp_sweep (include/sgp_hip.h)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_libs = {}


def library_path(variant=None) -> str:
    return _build.lib_path(variant)


def load(build_if_missing: bool = True, variant=None):
    """Load csrc/libsgp_hip.so -- or a variant library of the same ABI, see _build.VARIANTS -- building it first if hipcc is
    available.  Raises if it cannot."""
    if variant is None:
        variant = os.environ.get("SGP_LIB_VARIANT") or None      # diagnostics: run any script on a variant library (tools/step_trace.py)
    if variant in _libs:
        return _libs[variant]
    # PyTorch (used for streams / torch.distributed around this library) bundles its own libamdhip64.so.7.
    # A process must hold ONE HIP runtime: import torch first when it is installed so that this library's
    # DT_NEEDED libamdhip64.so.7 resolves to the copy torch already loaded (loading the system runtime
    # first leaves torch unable to see the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = _build.lib_path(variant)
    if not os.path.exists(path):
        if not build_if_missing:
            raise SGPError(f"{path} is missing: build it with `python -m gaussianprocessnode_amd._build`")
        _build.build(variant=variant)
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    lib.sgp_abi_version.restype = C.c_int
    lib.sgp_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.sgp_destroy.argtypes = [vp]
    lib.sgp_last_error.argtypes = [vp]
    lib.sgp_last_error.restype = C.c_char_p
    lib.sgp_set_inducing.argtypes = [vp, dp]
    lib.sgp_set_data.argtypes = [vp, dp, dp, dp, dp, C.c_int64, C.c_double]
    lib.sgp_set_output_cov_sum.argtypes = [vp, dp]
    lib.sgp_set_kernel.argtypes = [vp, C.c_double, dp, C.c_int32, C.c_double]
    lib.sgp_set_prior.argtypes = [vp, dp, dp, C.c_int32]
    lib.sgp_set_noise.argtypes = [vp, dp, C.c_double]
    for name in ("sgp_sweep_local", "sgp_sweep_finish", "sgp_sweep"):
        getattr(lib, name).argtypes = [vp, vp]
    lib.sgp_stats_layout.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.sgp_bind_stats.argtypes = [vp, vp]
    lib.sgp_get_posterior.argtypes = [vp, dp, dp, dp]
    lib.sgp_get_scalars.argtypes = [vp, dp]
    lib.sgp_get_stats.argtypes = [vp, dp, dp, dp]
    lib.sgp_get_kuu_chol.argtypes = [vp, dp]
    lib.sgp_get_wishart_invscale.argtypes = [vp, dp]
    lib.sgp_w_stats.argtypes = [vp, dp, dp, vp]
    lib.sgp_carry_posterior.argtypes = [vp, vp]
    lib.sgp_set_posterior.argtypes = [vp, dp, dp]
    lib.sgp_predict.argtypes = [vp, dp, C.c_int64, dp, dp]
    lib.sgp_theta_objective.argtypes = [vp, dp, dp]
    lib.sgp_kernelmatrix.argtypes = [C.c_int32, dp, C.c_int64, dp, C.c_int64, C.c_int32, C.c_double, dp, C.c_int32, dp]
    lib.sgp_potrf.argtypes = [C.c_int32, dp, C.c_int32, dp]
    lib.sgp_potri.argtypes = [C.c_int32, dp, C.c_int32, dp]
    lib.sgp_get_timestamps.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.sgp_get_phase_totals.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]
    lib.sgp_time_kernel.argtypes = [vp, C.c_int32, C.c_int32, vp, dp]
    lib.sgp_get_chain_trace.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
    lib.sgp_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.sgp_use_rccl.argtypes = [vp, vp]
    lib.sgp_measure_sclk_mhz.argtypes = [C.c_int32, dp]
    lib.sgp_train_begin.argtypes = [vp, dp, dp, C.c_int64, dp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.sgp_train_step.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32]
    lib.sgp_train_end.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    lib.sgp_get_step_trace.argtypes = [C.POINTER(C.c_int64)]
    lib.sgp_measure_clocks.argtypes = [C.c_int32, dp]
    lib.sgp_overlap_plan.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.sgp_get_sweep_trace.argtypes = [C.POINTER(C.c_int64)]
    lib.sgp_train_likelihood.argtypes = [vp, C.c_int32, C.c_double, C.c_double]
    lib.sgp_train_get_gamma.argtypes = [vp, dp]
    lib.sgp_wait.argtypes = [vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name != "sgp_last_error":
            fn.restype = C.c_int
    _libs[variant] = lib
    return lib


def as_f64(a, shape=None):
    """C-contiguous float64 view/copy (the memory layout the ABI expects: points along axis 0)."""
    arr = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and arr.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {arr.shape}")
    return arr


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def check(rc: int, handle=None, what: str = "", lib=None):
    if rc == 0:
        return
    lib = lib or load()
    msg = lib.sgp_last_error(handle)
    msg = msg.decode() if msg else ""
    if rc > 0:
        raise PosDefException(rc, f"{what} {msg}")
    raise SGPError(f"{what} failed with status {rc}: {msg}")
