"""`SGPDevice`: one rank's resident sparse-GP state on one MI355X, a thin object over the C ABI.

Array conventions follow the reference's Julia memory layout through NumPy C-order:
points `X` are (N, D) (= column-major D x N), `Xu` is (M, D), symmetric matrices are (M, M).
Nothing here computes on the CPU: every method is a call into csrc/libsgp_hip.so.
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


@dataclass
class SweepScalars:
    sum_I1: float
    sum_I2: float
    energy: float
    info_kuu: int
    info_lambda: int
    logdet_kuu: float
    logdet_lambda: float


# Handles that are still open when the interpreter starts to shut down are closed HERE -- an atexit handler runs at the very
# beginning of finalisation, while ctypes, the HIP runtime, torch and a profiler's tool library are all still alive -- and not
# from `__del__` during module teardown or, worse, never (a hook closure that references its engine keeps the handle in a
# reference cycle): a live handle with a registered ctypes hook at exit ended a rocprofv3-profiled run in SIGSEGV inside
# __cxa_finalize (round 3, tools/hooked_train.py).  The library keeps an exit handler of its own for C / Julia callers.
_live = weakref.WeakSet()


@atexit.register
def _close_all_at_exit():
    for dev in list(_live):
        try:
            dev.close()
        except Exception:                                  # pragma: no cover
            pass


class SGPDevice:
    """Owns the device buffers for (n_max points, M inducing points, D dims, d_out outputs)."""

    def __init__(self, n_max: int, m: int, d: int, d_out: int = 1, device: int = 0, use_graph: bool = False,
                 keep_kuf: bool = False, persistent_chain: bool = False):
        # (the round-2 persistent factorisation launch lives in a variant library of the same ABI, see _build.VARIANTS)
        self._lib = _lib.load(variant="chain" if persistent_chain else None)
        self._h = C.c_void_p()
        flags = ((_lib.SGP_FLAG_GRAPH if use_graph else 0) | (_lib.SGP_FLAG_KEEP_KUF if keep_kuf else 0)
                 | (_lib.SGP_FLAG_PERSISTENT_CHAIN if persistent_chain else 0))
        cfg = _lib.Config(n_max=int(n_max), m=int(m), d=int(d), d_out=int(d_out), device=int(device), flags=flags)
        check(self._lib.sgp_create(C.byref(cfg), C.byref(self._h)), None, "sgp_create", lib=self._lib)
        self.n_max, self.M, self.D, self.d_out, self.device = int(n_max), int(m), int(d), int(d_out), int(device)
        self.Q = self.M * self.d_out
        self.n = 0
        self._allreduce_cb = None
        _live.add(self)

    def _check(self, rc: int, what: str):
        check(rc, self._h, what, lib=self._lib)

    # ---- lifetime
    def close(self):
        """Destroy the handle (idempotent).  The all-reduce hook goes first: the library must never call a trampoline that is
        about to be freed, and `sgp_destroy` waits for whatever is still queued."""
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_allreduce_cb", None) is not None:
                self._lib.sgp_set_allreduce(self._h, _lib.ALLREDUCE_FN(0), None)
            self._lib.sgp_destroy(self._h)
            self._h = C.c_void_p()
            self._allreduce_cb = None
        _live.discard(self)

    def __del__(self):
        # not while the interpreter is finalising: modules this would call into may be half torn down -- the atexit handler
        # above has closed every live handle before that point
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- inputs
    def set_inducing(self, Xu):
        Xu = as_f64(np.reshape(Xu, (self.M, self.D)))
        self._check(self._lib.sgp_set_inducing(self._h, ptr(Xu)), "sgp_set_inducing")

    def set_data(self, X, y_mean, y_var=None, weights=None, n_nodes: Optional[float] = None):
        X = as_f64(np.reshape(X, (-1, self.D)))
        n = X.shape[0]
        # y: (n,) or (n, d_out) -> column-major n x d_out == C-order (d_out, n)
        y = np.asarray(y_mean, dtype=np.float64).reshape(n, self.d_out)
        y_cm = as_f64(y.T)
        yv = None if y_var is None else as_f64(np.reshape(y_var, (n,)))
        w = None if weights is None else as_f64(np.reshape(weights, (n,)))
        self._check(self._lib.sgp_set_data(self._h, ptr(X), ptr(y_cm), ptr(yv), ptr(w), n,
                                     float(-1.0 if n_nodes is None else n_nodes)), "sgp_set_data")
        self.n = n

    def set_output_cov_sum(self, S):
        S = as_f64(np.reshape(S, (self.d_out, self.d_out)))
        self._check(self._lib.sgp_set_output_cov_sum(self._h, ptr(as_f64(S.T))), "sgp_set_output_cov_sum")

    def set_kernel(self, sigma2: float, ell, jitter: float = 0.0):
        ell = as_f64(np.atleast_1d(ell))
        self._check(self._lib.sgp_set_kernel(self._h, float(sigma2), ptr(ell), int(ell.size), float(jitter)), "sgp_set_kernel")
        self._n_ell = int(ell.size)          # the C side writes 1 + n_ell gradient entries (sgp_theta_objective)

    def set_prior_meancov(self, mu0, Sigma0):
        mu0 = as_f64(np.reshape(mu0, (self.Q,)))
        S0 = as_f64(np.reshape(Sigma0, (self.Q, self.Q)))
        self._check(self._lib.sgp_set_prior(self._h, ptr(mu0), ptr(S0), 0), "sgp_set_prior")

    def set_prior_precision(self, xi0, Lambda0):
        xi0 = as_f64(np.reshape(xi0, (self.Q,)))
        L0 = as_f64(np.reshape(Lambda0, (self.Q, self.Q)))
        self._check(self._lib.sgp_set_prior(self._h, ptr(xi0), ptr(L0), 1), "sgp_set_prior")

    def set_prior_isotropic(self, variance: float):
        v = as_f64([variance])
        self._check(self._lib.sgp_set_prior(self._h, None, ptr(v), 2), "sgp_set_prior")

    def set_noise(self, W, E_log_w: Optional[float] = None):
        W = as_f64(np.reshape(W, (self.d_out, self.d_out)))
        if E_log_w is None:
            E_log_w = float(np.log(W[0, 0])) if self.d_out == 1 else float(np.linalg.slogdet(W)[1])
        self._check(self._lib.sgp_set_noise(self._h, ptr(as_f64(W.T)), float(E_log_w)), "sgp_set_noise")

    # ---- sweep
    def sweep_local(self, stream: int = 0):
        self._check(self._lib.sgp_sweep_local(self._h, C.c_void_p(stream)), "sgp_sweep_local")

    def sweep_finish(self, stream: int = 0):
        self._check(self._lib.sgp_sweep_finish(self._h, C.c_void_p(stream)), "sgp_sweep_finish")

    def set_allreduce(self, fn):
        """Install (fn = None: remove) the multi-GPU exchange step of `sweep`: fn(stats_dev_ptr, count, stream) must enqueue an
        in-place sum-all-reduce of `count` doubles on `stream` (include/sgp_hip.h, sgp_set_allreduce)."""
        if fn is None:
            self._allreduce_cb = None
            self._check(self._lib.sgp_set_allreduce(self._h, _lib.ALLREDUCE_FN(0), None), "sgp_set_allreduce")
            return

        def hook(ctx, buf, count, stream):
            try:
                fn(int(buf), int(count), int(stream or 0))
                return 0
            except Exception:                              # pragma: no cover  (reported through the status code)
                import traceback
                traceback.print_exc()
                return 1
        self._allreduce_cb = _lib.ALLREDUCE_FN(hook)        # keep the trampoline alive as long as it is installed
        self._check(self._lib.sgp_set_allreduce(self._h, self._allreduce_cb, None), "sgp_set_allreduce")

    def use_rccl(self, comm_ptr: int):
        """All-reduce with RCCL on an ncclComm_t the host program created (sgp_use_rccl)."""
        self._check(self._lib.sgp_use_rccl(self._h, C.c_void_p(comm_ptr)), "sgp_use_rccl")

    def sweep(self, stream: int = 0):
        self._check(self._lib.sgp_sweep(self._h, C.c_void_p(stream)), "sgp_sweep")

    def wait(self):
        """Returns when everything this handle has enqueued has finished (polled, then blocking: sgp_wait)."""
        self._check(self._lib.sgp_wait(self._h), "sgp_wait")

    def stats_layout(self):
        p, cnt, mp = C.c_void_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.sgp_stats_layout(self._h, C.byref(p), C.byref(cnt), C.byref(mp)), "sgp_stats_layout")
        return p.value, cnt.value, mp.value

    def bind_stats(self, dev_ptr: int):
        self._check(self._lib.sgp_bind_stats(self._h, C.c_void_p(dev_ptr)), "sgp_bind_stats")

    # ---- results
    def posterior(self, want_cov: bool = True, want_uv: bool = True):
        mu = np.empty(self.Q)
        Sig = np.empty((self.Q, self.Q)) if want_cov else None
        Uv = np.empty((self.Q, self.Q)) if want_uv else None
        self._check(self._lib.sgp_get_posterior(self._h, ptr(mu), ptr(Sig), ptr(Uv)), "sgp_get_posterior")
        # column-major upper-triangular Uv arrives as the C-order transpose
        return mu, Sig, (None if Uv is None else Uv.T.copy())

    def scalars(self) -> SweepScalars:
        out = np.empty(_lib.SGP_R_COUNT)
        self._check(self._lib.sgp_get_scalars(self._h, ptr(out)), "sgp_get_scalars")
        return SweepScalars(out[0], out[1], out[2], int(out[3]), int(out[4]), out[6], out[7])

    def stats(self):
        Psi2 = np.empty((self.M, self.M))
        B = np.empty((self.d_out, self.M))
        sc = np.empty(_lib.SGP_S_COUNT)
        self._check(self._lib.sgp_get_stats(self._h, ptr(Psi2), ptr(B), ptr(sc)), "sgp_get_stats")
        return Psi2, B.T.copy(), sc

    def kuu_chol(self):
        L = np.empty((self.M, self.M))
        self._check(self._lib.sgp_get_kuu_chol(self._h, ptr(L)), "sgp_get_kuu_chol")
        return L.T.copy()          # column-major lower -> C-order array holding L

    def wishart_invscale(self):
        S = np.empty((self.d_out, self.d_out))
        self._check(self._lib.sgp_get_wishart_invscale(self._h, ptr(S)), "sgp_get_wishart_invscale")
        return S.T.copy()

    def w_stats(self):
        I1, I2 = np.empty(self.n), np.empty(self.n)
        self._check(self._lib.sgp_w_stats(self._h, ptr(I1), ptr(I2), None), "sgp_w_stats")
        return I1, I2

    def predict(self, Xstar, mu_v=None):
        Xs = as_f64(np.reshape(Xstar, (-1, self.D)))
        ns = Xs.shape[0]
        out = np.empty((self.d_out, ns))
        mu = None if mu_v is None else as_f64(np.reshape(mu_v, (self.Q,)))
        self._check(self._lib.sgp_predict(self._h, ptr(Xs), ns, ptr(mu), ptr(out)), "sgp_predict")
        return out[0] if self.d_out == 1 else out.T.copy()

    def set_posterior(self, mu_v, Uv):
        """Install an external q(v) (mean and Uv = chol(Sigma_v + mu mu').U) for the per-point outputs (`w_stats`)."""
        mu = as_f64(np.reshape(mu_v, (self.Q,)))
        U = as_f64(np.asarray(Uv, dtype=np.float64).reshape(self.Q, self.Q).T)     # row-major Uv^T == column-major Uv
        self._check(self._lib.sgp_set_posterior(self._h, ptr(mu), ptr(U)), "sgp_set_posterior")

    def carry_posterior(self, stream: int = 0):
        """prior <- posterior of the last sweep, on the device (the minibatch carry, regression_kin40k.ipynb:205-212)."""
        self._check(self._lib.sgp_carry_posterior(self._h, C.c_void_p(stream)), "sgp_carry_posterior")

    def theta_objective(self, want_grad: bool = False, n_ell: Optional[int] = None):
        """neg_log_backwardmess_fast at the current kernel with q(v) fixed at the last sweep; optionally its gradient
        w.r.t. (sigma2, ell...)."""
        v = C.c_double()
        have = getattr(self, "_n_ell", None)
        if want_grad and have is None:
            raise ValueError("theta_objective: call set_kernel first")
        if n_ell is not None and have is not None and int(n_ell) != have:
            raise ValueError(f"theta_objective: n_ell={n_ell} but the kernel was set with {have} lengthscale(s)")
        g = np.empty(1 + have) if want_grad else None
        self._check(self._lib.sgp_theta_objective(self._h, C.cast(C.byref(v), C.POINTER(C.c_double)), ptr(g)), "sgp_theta_objective")
        return (v.value, g) if want_grad else v.value

    # -- device-paced minibatch training (sgp_train_*) -----------------------------------------------------------------
    def train_begin(self, X, y, theta_raw, *, jitter: float = 0.0, eta: float = 1e-3, beta=(0.9, 0.999), eps: float = 1e-8,
                    likelihood=None, gamma=(0.01, 0.01)):
        """Upload the training set (X: N x D, one point per row; y: N) and the raw (pre-softplus) parameters, reset the
        AdaMax state: the loop of `PerformInference` (experiments/regression_kin40k.ipynb:196-230) then runs as
        `train_step` calls that only enqueue."""
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64).reshape(-1, self.D))
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if len(y) != X.shape[0]:
            raise ValueError("train_begin: X and y disagree on the number of points")
        th = np.ascontiguousarray(np.asarray(theta_raw, dtype=np.float64).reshape(-1))
        n_ell = th.size - 1
        self._check(self._lib.sgp_train_begin(self._h, ptr(X), ptr(y), len(y), ptr(th), n_ell, float(jitter), float(eta),
                                        float(beta[0]), float(beta[1]), float(eps)), "sgp_train_begin")
        self._n_ell = n_ell
        if likelihood == "probit":
            # classification: y are labels in {0, 1}, q(w) = Gamma(*gamma) carried over the minibatches (sgp_train_likelihood)
            self._check(self._lib.sgp_train_likelihood(self._h, 1, float(gamma[0]), float(gamma[1])), "sgp_train_likelihood")
        elif likelihood not in (None, "gaussian"):
            raise ValueError(f"train_begin: unknown likelihood {likelihood!r}")

    def train_gamma(self):
        """(shape, rate) of q(w) after a classification run (sgp_train_get_gamma)."""
        ab = np.empty(2)
        self._check(self._lib.sgp_train_get_gamma(self._h, ptr(ab)), "sgp_train_get_gamma")
        return float(ab[0]), float(ab[1])

    def train_step(self, offset: int, n: int, learn: bool = True, reset_prior: bool = False):
        """One minibatch [offset, offset + n): sweep, carry, gradient, optimiser step.  Asynchronous.  reset_prior puts
        the isotropic prior of `set_prior_isotropic` back first (the per-epoch reset of the notebooks)."""
        self._check(self._lib.sgp_train_step(self._h, int(offset), int(n), (1 if learn else 0) | (2 if reset_prior else 0)),
                    "sgp_train_step")

    def train_end(self):
        """Wait for the queued steps; returns (theta_raw, optimiser steps taken, minibatches skipped)."""
        th = np.empty(1 + self._n_ell)
        counts = (C.c_int64 * 2)()
        self._check(self._lib.sgp_train_end(self._h, ptr(th), counts), "sgp_train_end")
        return th, int(counts[0]), int(counts[1])

    def time_kernel(self, which: int, iters: int = 20, stream: int = 0) -> float:
        """Average launch duration (microseconds, HIP events) of the Gram or streaming-SYRK kernel."""
        v = C.c_double()
        self._check(self._lib.sgp_time_kernel(self._h, int(which), int(iters), C.c_void_p(stream),
                                        C.cast(C.byref(v), C.POINTER(C.c_double))), "sgp_time_kernel")
        return v.value

    def overlap_plan(self):
        """What the next `sweep()` will do (sgp_overlap_plan): [] = statistics first, then the Lambda chain; else one dict per
        statistics group of the overlapped sweep."""
        n = C.c_int32()
        info = (C.c_int32 * 64)()
        self._check(self._lib.sgp_overlap_plan(self._h, C.byref(n), info), "sgp_overlap_plan")
        keys = ("col_begin", "col_end", "tiles", "chunks", "points_per_chunk", "masked", "cus", "form_step")
        return [dict(zip(keys, info[8 * g:8 * g + 8])) for g in range(n.value)]

    def time_group(self, g: int, iters: int = 20) -> float:
        """Average duration (microseconds, HIP events on the group's own stream) of the SYRK launch of statistics group g."""
        return self.time_kernel(_lib.SGP_TIME_GROUP0 + int(g), iters)

    def phase_totals(self, reset: bool = False):
        """(average microseconds per sweep for every phase slot, number of sweeps counted) since the last reset."""
        tot = (C.c_int64 * _lib.SGP_T_COUNT)()
        cnt = C.c_int64()
        self._check(self._lib.sgp_get_phase_totals(self._h, tot, C.byref(cnt), int(reset)), "sgp_get_phase_totals")
        n = max(cnt.value, 1)
        return np.array(tot[:], dtype=np.float64) / 100.0 / n, cnt.value

    def timestamps(self):
        out = (C.c_int64 * (2 * _lib.SGP_T_COUNT))()
        self._check(self._lib.sgp_get_timestamps(self._h, out), "sgp_get_timestamps")
        return np.array(out[:], dtype=np.int64).reshape(_lib.SGP_T_COUNT, 2)


# ---- stand-alone building blocks -------------------------------------------------------------
def kernelmatrix(A, B, sigma2: float, ell, device: int = 0):
    """K(A, B) on the device: sigma2 * exp(-0.5 |(a-b)/ell|^2); A (na, D), B (nb, D) -> (na, nb)."""
    lib = _lib.load()
    A = as_f64(np.atleast_2d(A))
    B = as_f64(np.atleast_2d(B))
    ell = as_f64(np.atleast_1d(ell))
    K = np.empty((B.shape[0], A.shape[0]))            # column-major na x nb
    check(lib.sgp_kernelmatrix(device, ptr(A), A.shape[0], ptr(B), B.shape[0], A.shape[1], float(sigma2), ptr(ell),
                               int(ell.size), ptr(K)), None, "sgp_kernelmatrix")
    return K.T.copy()


def potrf(A, device: int = 0, variant=None):
    """Lower Cholesky factor on the device (fastcholesky(A).L)."""
    lib = _lib.load(variant=variant)
    A = as_f64(A)
    n = A.shape[0]
    L = np.empty((n, n))
    check(lib.sgp_potrf(device, ptr(as_f64(A.T)), n, ptr(L)), None, "sgp_potrf", lib=lib)
    return L.T.copy()


def potri(A, device: int = 0, variant=None):
    """Inverse of an SPD matrix through its Cholesky factor on the device (cholinv(A))."""
    lib = _lib.load(variant=variant)
    A = as_f64(A)
    n = A.shape[0]
    out = np.empty((n, n))
    check(lib.sgp_potri(device, ptr(as_f64(A.T)), n, ptr(out)), None, "sgp_potri", lib=lib)
    return out.T.copy()
