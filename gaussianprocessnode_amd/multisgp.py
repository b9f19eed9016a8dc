"""Host-side mirror of the `MultiSGP` factor node (GPnode/MultiSGPnode.jl): D_out outputs sharing one kernel, uncertain
(Gaussian) inputs handled by cubature, Wishart-distributed noise precision.

The reference evaluates, per time step, cubature Psi-statistics (5 Gram columns + 5 rank-1 M x M updates), a
`kron(W, Psi2)` message and a DM x DM Gaussian product.  Here the cubature points of ALL steps go to the device as weighted
data in one call; the summed statistics give q(v), the Wishart inverse scale and the average energy in one sweep
(eq. M).  Per-step rule functions are provided for interface parity (they run the same device
kernels on one step).  Everything numeric comes from `meta.engine` (C ABI); no CPU fallback.
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


def rule_in(*args, **kwargs):
    raise NotImplementedError("MultiSGP(:in) (GPnode/MultiSGPnode.jl:162-236) returns a log-pdf closure / Laplace fit "
                              "evaluated inside ReactiveMP; not on the device path yet (SURVEY.md §8 a14)")


def rule_theta(*args, **kwargs):
    raise NotImplementedError("MultiSGP(:theta) (GPnode/MultiSGPnode.jl:447-469) returns a log-pdf closure (SURVEY.md §8 f1)")
