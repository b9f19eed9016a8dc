"""Host-side mirror of the `MultiSGP` factor node (GPnode/MultiSGPnode.jl): D_out outputs sharing one kernel, uncertain
(Gaussian) inputs handled by cubature, Wishart-distributed noise precision.

The reference evaluates, per time step, cubature Psi-statistics (5 Gram columns + 5 rank-1 M x M updates), a
`kron(W, Psi2)` message and a DM x DM Gaussian product.  Here the cubature points of ALL steps go to the device as weighted
data in one call; the summed statistics give q(v), the Wishart inverse scale and the average energy in one sweep
(eq. M).  Per-step rule functions are provided for interface parity (they run the same device
kernels on one step).

What runs where: Gram matrices, Psi-statistics, every factorisation / inverse, q(v), the Wishart inverse scale, the energy
and the per-point quadratic forms come from `meta.engine` and the device building blocks (C ABI); there is no CPU
fallback for them.  What stays in NumPy is bookkeeping of a few d_out x d_out or M x M operands that are arguments of
those calls, not results of the node's algebra: cubature points and weights of q_in (cubature.py), `slogdet` of the
d_out x d_out mean(q_w) when q_w has no `mean_logdet` (:83), the block contraction S = sum_ij W_ij Rv[i][j] and
s = sum_d mu_v^(d) (mu_y' W)_d of `_second_moment_contraction`, a trace, and the Laplace step of `rule_in` (L-BFGS and a
finite-difference Hessian in d_in dimensions around the device-evaluated closure; the reference uses ForwardDiff).
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

from .distributions import (MvNormalMeanCovariance, MvNormalMeanPrecision, MvNormalWeightedMeanPrecision, PointMass,
                            WishartFast)
from .meta import MultiSGPMeta


class MultiSGP:
    """Node tag: `@node MultiSGP Stochastic [out, in, v, w, theta]` (GPnode/MultiSGPnode.jl:47-49)."""
    interfaces = ("out", "in", "v", "w", "theta")


def _mean_W(q_w):
    W = q_w.mean() if hasattr(q_w, "mean") else q_w
    return np.atleast_2d(np.asarray(W, dtype=np.float64))


def _cov_of(q):
    return None if isinstance(q, PointMass) else np.atleast_2d(np.asarray(q.cov(), dtype=np.float64))


def _engine(meta: MultiSGPMeta, n_points: int, d_out: int):
    Xu = np.asarray(meta.Xu, dtype=np.float64)
    M, D = Xu.shape
    eng = meta.engine
    if eng is None or eng.n_max < n_points or eng.d_out != d_out:
        from .device import SGPDevice
        if eng is not None:
            eng.close()
        eng = SGPDevice(max(n_points, 1), M, D, d_out, device=meta.device)
        eng.set_inducing(Xu)
        meta.engine = eng
    return eng


def _expand(meta: MultiSGPMeta, q_ins: Sequence, q_outs: Sequence):
    """All steps' cubature points as weighted data (approximate_kernel_expectation!, GPnode/MultiSGPnode.jl:11-35)."""
    pts, wts, ys, cov_sum = [], [], [], None
    for q_in, q_out in zip(q_ins, q_outs):
        if isinstance(q_in, PointMass):
            p, w = np.atleast_2d(np.asarray(q_in.mean(), dtype=np.float64)), np.ones(1)
        else:
            m, P = q_in.mean_cov()
            p, w = meta.method.points_weights(m, P)
        y = np.asarray(q_out.mean(), dtype=np.float64).ravel()
        pts.append(p)
        wts.append(w)
        ys.append(np.repeat(y[None, :], len(w), axis=0))
        c = _cov_of(q_out)
        if c is not None:
            cov_sum = c.copy() if cov_sum is None else cov_sum + c
    return np.concatenate(pts), np.concatenate(wts), np.concatenate(ys), cov_sum


def sweep(meta: MultiSGPMeta, q_outs: Sequence, q_ins: Sequence, q_w, q_theta: PointMass, prior, E_logdet_W=None):
    """One VMP update of q(v) from all steps' `:v` messages (GPnode/MultiSGPnode.jl:290-328) folded with the prior.
    Returns the marginal MvNormalMeanCovariance; the Wishart statistics and the energy are then available through
    `rule_w_summed` / `average_energy_summed`."""
    pts, wts, ys, cov_sum = _expand(meta, q_ins, q_outs)
    W = _mean_W(q_w)
    d_out = W.shape[0]
    eng = _engine(meta, len(wts), d_out)
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64)))
    eng.set_data(pts, ys, None, wts, n_nodes=len(q_ins))
    if cov_sum is not None:
        eng.set_output_cov_sum(cov_sum)
    eng.set_kernel(sigma2, ell, meta.jitter)
    if E_logdet_W is None:
        E_logdet_W = q_w.mean_logdet() if hasattr(q_w, "mean_logdet") else float(np.linalg.slogdet(W)[1])
    eng.set_noise(W, E_logdet_W)
    if isinstance(prior, MvNormalMeanCovariance):
        eng.set_prior_meancov(prior.m, prior.S)
    elif isinstance(prior, MvNormalWeightedMeanPrecision):
        eng.set_prior_precision(prior.xi, prior.W)
    elif isinstance(prior, MvNormalMeanPrecision):
        eng.set_prior_precision(prior.W @ prior.m, prior.W)
    else:
        raise TypeError(f"unsupported prior type {type(prior).__name__}")
    eng.sweep()
    mu, Sigma, _ = eng.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma)


def rule_w_summed(meta: MultiSGPMeta, prior_nu: float, prior_invscale, n_nodes: int) -> WishartFast:
    """q(W) = prior x all `:w` messages WishartFast(D + 2, I1_t + I2_t) (GPnode/MultiSGPnode.jl:367-444):
    inverse scales add, degrees of freedom nu0 + N."""
    S = meta.engine.wishart_invscale()
    return WishartFast(prior_nu + n_nodes, np.asarray(prior_invscale, dtype=np.float64) + S)


def average_energy_summed(meta: MultiSGPMeta) -> float:
    """Sum over the steps of @average_energy MultiSGP (GPnode/MultiSGPnode.jl:544-632)."""
    return meta.engine.scalars().energy


def rule_out(q_in, q_v, q_w, q_theta: PointMass, meta: MultiSGPMeta) -> MvNormalMeanPrecision:
    """@rule MultiSGP(:out) (GPnode/MultiSGPnode.jl:90-120): mean_d = Psi1 . mu_v^(d), precision mean(q_w)."""
    W = _mean_W(q_w)
    d_out = W.shape[0]
    if isinstance(q_in, PointMass):
        p, w = np.atleast_2d(np.asarray(q_in.mean(), dtype=np.float64)), np.ones(1)
    else:
        p, w = meta.method.points_weights(*q_in.mean_cov())
    eng = _engine(meta, len(w), d_out)
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64)))
    eng.set_kernel(sigma2, ell, meta.jitter)
    f = np.atleast_2d(eng.predict(p, np.asarray(q_v.mean(), dtype=np.float64)))      # (S, d_out)
    if f.shape[0] != len(w):
        f = f.T
    return MvNormalMeanPrecision(w @ f, W)


def rule_v(q_out, q_in, q_w, q_theta: PointMass, meta: MultiSGPMeta) -> MvNormalWeightedMeanPrecision:
    """@rule MultiSGP(:v) for ONE step (GPnode/MultiSGPnode.jl:290-328): the message itself, xi = vcat(Psi1 (mu_y' W)_d),
    Lambda = kron(W, Psi2), with Psi1 / Psi2 from the device statistics of that step."""
    W = _mean_W(q_w)
    pts, wts, ys, _ = _expand(meta, [q_in], [q_out])
    d_out = W.shape[0]
    eng = _engine(meta, len(wts), d_out)
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64)))
    eng.set_data(pts, np.ones((len(wts), d_out)), None, wts, n_nodes=1)
    eng.set_kernel(sigma2, ell, meta.jitter)
    eng.sweep_local()
    Psi2, B, _ = eng.stats()
    Psi1 = B[:, 0]
    row = np.asarray(q_out.mean(), dtype=np.float64).ravel() @ W                      # :307
    xi = np.concatenate([Psi1 * row[d] for d in range(d_out)])
    return MvNormalWeightedMeanPrecision(xi, np.kron(W, Psi2))                         # :306


def _aux_engine(meta: MultiSGPMeta, n: int):
    """A single-output device object for stand-alone closure evaluations (it replaces data and posterior, so it is not the
    engine that holds the last swept sequence)."""
    Xu = np.asarray(meta.Xu, dtype=np.float64)
    M, D = Xu.shape
    eng = getattr(meta, "_aux_engine", None)
    if eng is None or eng.n_max < n:
        from .device import SGPDevice
        if eng is not None:
            eng.close()
        eng = SGPDevice(max(n, 64), M, D, 1, device=meta.device, keep_kuf=True)
        eng.set_inducing(Xu)
        meta._aux_engine = eng
    return eng


def rule_in(q_out, q_v, q_w, q_theta: PointMass, meta: MultiSGPMeta, q_in=None):
    """@rule MultiSGP(:in) (GPnode/MultiSGPnode.jl:162-184 Gaussian output, :186-208 point-mass output): the log-pdf closure
        x -> -1/2 tr(W) I1(x) + s . k(x) - 1/2 k(x)' S k(x),      I1(x) = k(x,x) - k' Kuu^-1 k,
        s = sum_d mu_v^(d) (mu_y' W)_d  (sum_diagonal_M),   S = sum_ij W_ij Rv_blk[i][j]  (create_blockmatrix), Rv = Sigma_v + mu mu'.
    Both quadratic forms are per-point quantities the device already produces: with the pseudo-posterior (mean s, factor
    chol(S).U) and pseudo-observations y = 1, sgp_w_stats returns I1(x) and I2(x) = 1 - 2 s.k + k' S k, so the closure is
    -1/2 tr(W) I1 - 1/2 (I2 - 1) -- one device pass for any number of inputs (K_uu chain, K_uf, the two quadratic forms).
    K_uu^-1 is the device's own at (theta, meta.jitter); the reference reads the copy stored in its meta (:168), made the same way.
    With `q_in` given and point-mass q_out / q_w (:210-236) the closure is fitted by a Gaussian at its mode (Laplace): see
    `rule_in_laplace`."""
    if q_in is not None:
        return rule_in_laplace(q_out, q_in, q_v, q_w, q_theta, meta)
    from .unisgp import LogPdfClosure
    from .device import potrf
    Xu = np.asarray(meta.Xu, dtype=np.float64)
    M, D_in = Xu.shape
    W = _mean_W(q_w)
    s_vec, S = _second_moment_contraction(q_out, q_v, W, M)                             # :176-179
    US = potrf(S, meta.device).T                                                        # upper factor: |US k|^2 = k' S k
    trW = float(np.trace(W))
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64)))

    def log_backwardmess(x):
        X = np.asarray(x, dtype=np.float64).reshape(-1, D_in)
        eng = _aux_engine(meta, len(X))
        eng.set_data(X, np.ones(len(X)), None)
        eng.set_kernel(sigma2, ell, meta.jitter)
        eng.sweep_local()
        eng.set_posterior(s_vec, US)
        I1, I2 = eng.w_stats()
        val = -0.5 * trW * I1 - 0.5 * (I2 - 1.0)                                        # :181
        return float(val[0]) if np.ndim(x) == 1 else val
    return LogPdfClosure(log_backwardmess, multivariate=True)


def rule_in_laplace(q_out, q_in, q_v, q_w, q_theta: PointMass, meta: MultiSGPMeta, iterations: int = 20):
    """@rule MultiSGP(:in) with q_out::PointMass, q_in Gaussian, q_w::PointMass (GPnode/MultiSGPnode.jl:210-236): minimise the
    negative closure from mean(q_in) with L-BFGS (20 iterations, :229) and return N(m_z, W_z^-1) in weighted-mean form, W_z
    the Hessian at the minimiser (:231-233).  The reference differentiates with ForwardDiff / Zygote; here gradient and
    Hessian are central differences of the device-evaluated closure, every stencil one device pass."""
    from scipy.optimize import minimize
    closure = rule_in(q_out, q_v, q_w, q_theta, meta)
    x0 = np.asarray(q_in.mean(), dtype=np.float64).ravel()
    D_in = len(x0)
    h = 1e-5

    def neg_and_grad(x):
        pts = np.vstack([x] + [x + h * e for e in np.eye(D_in)] + [x - h * e for e in np.eye(D_in)])
        f = -np.asarray(closure.logpdf(pts))
        return float(f[0]), (f[1:1 + D_in] - f[1 + D_in:]) / (2 * h)
    res = minimize(neg_and_grad, x0, jac=True, method="L-BFGS-B", options={"maxiter": iterations})
    m_z = res.x
    hh = 1e-4
    E = np.eye(D_in)
    pts, idx = [m_z], {}
    for a in range(D_in):
        for b in range(a, D_in):
            idx[(a, b)] = len(pts)
            pts += [m_z + hh * (E[a] + E[b]), m_z + hh * (E[a] - E[b]), m_z - hh * (E[a] - E[b]), m_z - hh * (E[a] + E[b])]
    f = -np.asarray(closure.logpdf(np.vstack(pts)))
    W_z = np.zeros((D_in, D_in))
    for (a, b), k in idx.items():
        W_z[a, b] = W_z[b, a] = (f[k] - f[k + 1] - f[k + 2] + f[k + 3]) / (4 * hh * hh)
    return MvNormalWeightedMeanPrecision(W_z @ m_z, W_z)                                # :235


def _second_moment_contraction(q_out, q_v, W, M):
    """s = sum_d mu_v^(d) (mu_y' W)_d and S = sum_ij W_ij Rv_blk[i][j] -- what the :in and :theta closures keep of q(v)."""
    mu_y = np.asarray(q_out.mean(), dtype=np.float64).ravel()
    d_out = len(mu_y)
    mu_v, Sigma_v = q_v.mean_cov()
    mu_v = np.asarray(mu_v, dtype=np.float64).ravel()
    Rv = np.asarray(Sigma_v, dtype=np.float64) + np.outer(mu_v, mu_v)
    row = mu_y @ W
    s_vec = sum(mu_v[d * M:(d + 1) * M] * row[d] for d in range(d_out))
    S = sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * W[i, j] for i in range(d_out) for j in range(d_out))
    return s_vec, 0.5 * (S + S.T)


def rule_theta(q_out, q_in, q_v, q_w, meta: MultiSGPMeta):
    """@rule MultiSGP(:theta) (GPnode/MultiSGPnode.jl:447-466): theta -> -1/2 tr(W) (Psi0 - tr(Kuu^-1 Psi2')) + Psi1 . s
    - 1/2 tr(Psi2' S), Psi2' = Psi2 + 1e-7 I (:458), Kuu(theta) without jitter (:455), Psi by meta.method's cubature over q_in.
    Per cubature point this is the :in closure, so one device pass per theta (K_uu chain at theta, K_uf for the points, the
    two per-point quadratic forms); the 1e-7 I term adds 1e-7 (tr(W) tr(Kuu^-1) - tr(S)) / 2, with Kuu^-1 from the device
    (sgp_kernelmatrix + sgp_potri); the host only takes its trace."""
    from .unisgp import LogPdfClosure
    from .device import kernelmatrix, potri, potrf
    if meta.method is None:
        raise ValueError("MultiSGP(:theta) needs meta.method (a cubature rule)")
    Xu = np.asarray(meta.Xu, dtype=np.float64)
    M = Xu.shape[0]
    W = _mean_W(q_w)
    s_vec, S = _second_moment_contraction(q_out, q_v, W, M)
    US = potrf(S, meta.device).T
    trW, trS = float(np.trace(W)), float(np.trace(S))
    m_in, P_in = q_in.mean_cov()
    pts, wts = meta.method.points_weights(m_in, P_in)
    pts, wts = np.atleast_2d(np.asarray(pts, dtype=np.float64)), np.asarray(wts, dtype=np.float64)

    def log_backwardmess(theta):
        sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(theta, dtype=np.float64)))
        eng = _aux_engine(meta, len(pts))
        eng.set_data(pts, np.ones(len(pts)), None)
        eng.set_kernel(sigma2, ell, 0.0)
        eng.sweep_local()
        eng.set_posterior(s_vec, US)
        I1, I2 = eng.w_stats()
        tr_kinv = float(np.trace(potri(kernelmatrix(Xu, Xu, sigma2, ell, meta.device), meta.device)))
        return float(wts @ (-0.5 * trW * I1 - 0.5 * (I2 - 1.0)) + 0.5e-7 * (trW * tr_kinv - trS))
    return LogPdfClosure(log_backwardmess, multivariate=True)
