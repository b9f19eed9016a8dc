"""ctypes access to oracle/libsgp_oracle.so (the per-point C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsgp_oracle.so")


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "sgp_oracle.c")):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    lib = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    lib.oracle_vmp_sweep_perpoint.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp, C.c_long, C.c_double, dp, C.c_int,
                                              C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp]
    lib.oracle_vmp_sweep_perpoint.restype = C.c_int
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def vmp_sweep_perpoint(Xu, X, y, y_var, sigma2, ell, jitter, w_bar, E_logw, mu0, Sigma0, want_points=False):
    lib = load()
    Xu = np.ascontiguousarray(Xu, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, Xu.shape[1])
    y = np.ascontiguousarray(y, dtype=np.float64)
    yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
    ell = np.ascontiguousarray(np.atleast_1d(ell), dtype=np.float64)
    M, D = Xu.shape
    N = X.shape[0]
    mu0 = np.ascontiguousarray(mu0, dtype=np.float64)
    S0 = np.ascontiguousarray(Sigma0, dtype=np.float64)
    mu, Sig, Uv, out = np.empty(M), np.empty((M, M)), np.empty((M, M)), np.empty(3)
    I1 = np.empty(N) if want_points else None
    I2 = np.empty(N) if want_points else None
    rc = lib.oracle_vmp_sweep_perpoint(_p(Xu), M, D, _p(X), _p(y), _p(yv), N, sigma2, _p(ell), ell.size, jitter, w_bar,
                                       E_logw, _p(mu0), _p(S0), _p(mu), _p(Sig), _p(Uv), _p(out), _p(I1), _p(I2))
    if rc != 0:
        raise ArithmeticError(f"per-point oracle: Cholesky failed at minor {rc}")
    return dict(mu_v=mu, Sigma_v=Sig, Uv=Uv.T.copy(), sum_I1=out[0], sum_I2=out[1], energy=out[2], I1=I1, I2=I2)
