"""Test doubles: CPU engines with the `SGPDevice` / distributed-engine interface, computing through the oracle.
They exist so that the HOST logic (counter hook, packing, sharding, the all-reduce path under gloo) can be
exercised in the GPU-less container.  They live under tests/ on purpose: the product has no CPU fallback."""
import math

import numpy as np

from gaussianprocessnode_amd.device import SweepScalars
from gaussianprocessnode_amd.distributed import pack_stats, stats_count, unpack_stats
from oracle import sgp_oracle as O


class OracleDevice:
    """Same methods as gaussianprocessnode_amd.SGPDevice, NumPy inside."""

    def __init__(self, n_max, m, d, d_out=1, **kw):
        self.n_max, self.M, self.D, self.d_out = n_max, m, d, d_out
        self.Lambda0, self.xi0 = np.eye(m), np.zeros(m)
        self.jitter, self.w, self.E_logw = 0.0, 1.0, 0.0
        self.calls = []

    def close(self):
        pass

    def set_inducing(self, Xu):
        self.Xu = np.asarray(Xu, dtype=np.float64).reshape(self.M, self.D)

    def set_data(self, X, y, y_var=None, weights=None, n_nodes=None):
        self.X = np.asarray(X, dtype=np.float64).reshape(-1, self.D)
        self.y, self.vy = np.asarray(y, dtype=np.float64), y_var
        self.n = len(self.X)
        self.calls.append(("set_data", self.n))

    def set_kernel(self, sigma2, ell, jitter=0.0):
        self.s2, self.ell, self.jitter = sigma2, np.asarray(ell, dtype=np.float64), jitter

    def set_prior_meancov(self, mu0, S0):
        self.Lambda0 = O.cholinv(S0)
        self.xi0 = self.Lambda0 @ mu0

    def set_prior_precision(self, xi0, L0):
        self.Lambda0, self.xi0 = np.asarray(L0), np.asarray(xi0)

    def set_prior_isotropic(self, var):
        self.Lambda0, self.xi0 = np.eye(self.M) / var, np.zeros(self.M)

    def set_noise(self, W, E_log_w=None):
        self.w = float(np.asarray(W).ravel()[0])
        self.E_logw = math.log(self.w) if E_log_w is None else E_log_w

    def set_allreduce_array(self, fn):
        """Test double of the C ABI's all-reduce hook (include/sgp_hip.h, sgp_set_allreduce): fn(array) sums a float64 NumPy array
        over the ranks in place.  With it the sweep reduces the packed statistics and `theta_objective` its value and gradient."""
        self._reduce = fn

    def sweep(self, stream=0):
        self.calls.append(("sweep", self.n))
        if getattr(self, "_reduce", None) is None:
            self.res = O.vmp_sweep(self.Xu, self.X, self.y, self.vy, self.s2, self.ell, self.w, E_logw=self.E_logw,
                                   jitter=self.jitter, Lambda0=self.Lambda0, xi0=self.xi0)
            return
        # data-sharded: local statistics in the device's packed layout -> hook -> replicated tail (sgp_sweep with a hook)
        if self.n > 0:
            st = O.suff_stats(self.Xu, self.X, self.y, self.vy, self.s2, self.ell)
            buf = pack_stats(st.Psi2, st.b, float(st.s_yy[0, 0]), st.s_kk / self.s2, st.n)
        else:
            buf = np.zeros(stats_count(self.M, 1))
        self._reduce(buf)
        Psi2, B, s_yy, s_w, n, _ = unpack_stats(buf, self.M, 1)
        st = O.SuffStats(Psi2, B, np.array([[s_yy]]), self.s2 * s_w, n)
        self.res = O.vmp_sweep(self.Xu, None, None, None, self.s2, self.ell, self.w, E_logw=self.E_logw, jitter=self.jitter,
                               Lambda0=self.Lambda0, xi0=self.xi0, stats=st)

    def carry_posterior(self, stream=0):
        """prior <- posterior in natural form (sgp_carry_posterior)"""
        self.Lambda0 = self.Lambda0 + self.w * self.res.stats.Psi2
        self.xi0 = self.xi0 + self.w * np.ravel(self.res.stats.b)

    def theta_objective(self, want_grad=False, n_ell=None):
        """neg_log_backwardmess_fast at the current kernel with q(v) of the last sweep (helper_functions/derivative_helper.jl:23-39);
        gradient by central differences of this shard's terms, summed over the ranks when a hook is installed."""
        r = self.res
        ell = np.atleast_1d(self.ell).astype(np.float64)

        def f(p):
            if self.n == 0:
                return 0.0
            return O.theta_objective(self.Xu, self.X, self.y, p[0], p[1:], r.mu_v, r.Uv, self.w, jitter=self.jitter)
        p0 = np.concatenate([[self.s2], ell])
        out = np.array([f(p0)] + ([(f(p0 + 1e-6 * e) - f(p0 - 1e-6 * e)) / 2e-6 for e in np.eye(len(p0))] if want_grad else []))
        if getattr(self, "_reduce", None) is not None:
            self._reduce(out)
        return (float(out[0]), out[1:].copy()) if want_grad else float(out[0])

    def posterior(self, want_cov=True, want_uv=True):
        return self.res.mu_v, self.res.Sigma_v, self.res.Uv

    def kuu_chol(self):
        return self.res.KuuL

    def scalars(self):
        r = self.res
        return SweepScalars(r.sum_I1, r.sum_I2, r.energy, 0, 0, 0.0, 0.0)

    def w_stats(self):
        return O.w_stats_perpoint(self.Xu, self.X, self.y, self.vy, self.s2, self.ell, self.res.KuuL, self.res.mu_v,
                                  self.res.Uv)

    def predict(self, Xstar, mu_v=None):
        return O.predict_mean(self.Xu, Xstar, self.res.mu_v if mu_v is None else mu_v, self.s2, self.ell)


class OracleShardEngine:
    """Engine for gaussianprocessnode_amd.distributed.ShardedSweep on the CPU (gloo tests): local statistics of this
    rank's shard packed in the device layout into a torch CPU tensor, replicated tail from the reduced buffer."""

    def __init__(self, Xu, X, y, s2, ell, w, prior_var, jitter=0.0):
        import torch
        self.Xu, self.X, self.y = Xu, X, y
        self.s2, self.ell, self.w, self.prior_var, self.jitter = s2, ell, w, prior_var, jitter
        self.M = Xu.shape[0]
        self.stats = torch.zeros(stats_count(self.M, 1), dtype=torch.float64)

    def sweep_local(self):
        import torch
        st = O.suff_stats(self.Xu, self.X, self.y, None, self.s2, self.ell)
        buf = pack_stats(st.Psi2, st.b, float(st.s_yy[0, 0]), st.s_kk / self.s2, st.n)
        self.stats.copy_(torch.from_numpy(buf))

    def sweep_finish(self):
        Psi2, B, s_yy, s_w, n, _ = unpack_stats(self.stats.numpy(), self.M, 1)
        st = O.SuffStats(Psi2, B, np.array([[s_yy]]), self.s2 * s_w, n)
        self.res = O.vmp_sweep(self.Xu, None, None, None, self.s2, self.ell, self.w, jitter=self.jitter,
                               Lambda0=np.eye(self.M) / self.prior_var, xi0=np.zeros(self.M), stats=st)

    # the C ABI's form of the N > 1 path (include/sgp_hip.h, sgp_set_allreduce): the exchange step is a hook INSIDE the sweep
    def install_allreduce(self, reduce_fn):
        self._hook = reduce_fn
        self.hook_calls = 0

    def sweep(self):
        self.sweep_local()
        if getattr(self, "_hook", None) is not None:
            self._hook(self.stats)             # (HipEngine's form: the hook is handed the tensor to reduce)
            self.hook_calls += 1
        self.sweep_finish()

    def theta_objective_local(self, n_ell=None):
        """This shard's terms of neg_log_backwardmess_fast at the replicated q(v), gradient by central differences."""
        r = self.res
        f = lambda p: O.theta_objective(self.Xu, self.X, self.y, p[0], p[1:], r.mu_v, r.Uv, self.w, jitter=self.jitter)
        p0 = np.concatenate([[self.s2], self.ell])
        g = np.array([(f(p0 + 1e-6 * e) - f(p0 - 1e-6 * e)) / 2e-6 for e in np.eye(len(p0))])
        return f(p0), g
