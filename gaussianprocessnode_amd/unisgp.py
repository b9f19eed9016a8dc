"""Host-side mirror of the `UniSGP` factor node (GPnode/UniSGPnode.jl): same rule names, argument meaning and
error behaviour, batched on the device.

The reference's rules run once per data point and the N-fold Gaussian product folds N rank-1 M x M messages
(GPnode/UniSGPnode.jl:62-73,144-173).  Here the per-point `:v` rule returns an O(1) token and the product runs
ONE device sweep when the N-th token is folded -- the same `counter == N` hook the reference uses to refresh
`meta.Uv` (:64-71).  The PointMass-input rules (`q_in::PointMass`) are the hot path; of the uncertain-input variants the
`:v` (:125-140) and `:out` (:85-93) rules run on the device as cubature-weighted data.  The per-node clamped `:w` /
average-energy variants (:177-192,290-313), `:in` and the `:theta` closures (:242-287) raise NotImplementedError
(next).

Nothing here computes the node's mathematics on the CPU: every number comes from `meta.engine`
(`SGPDevice`, the C ABI).  Without the HIP library and a gfx950 GPU the first sweep raises.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .distributions import (GammaShapeRate, MvNormalMeanCovariance, MvNormalMeanPrecision,
                            MvNormalWeightedMeanPrecision, NormalMeanPrecision, NormalMeanVariance, PointMass)
from .meta import UniSGPMeta

LOG2PI = math.log(2.0 * math.pi)


class UniSGP:
    """Node tag: `@node UniSGP Stochastic [out, in, v, w, theta]` (GPnode/UniSGPnode.jl:76-78)."""
    interfaces = ("out", "in", "v", "w", "theta")


@dataclass
class BufferUniSGP:
    """Message wrapper of the `:v` rule (GPnode/UniSGPnode.jl:56-59).  `qv` is an O(1) token here (the index of the
    point in the pending batch), not an M x M message."""
    qv: object
    meta: UniSGPMeta


@dataclass
class PendingMarginal:
    """What the product returns before the N-th message has been folded (the reference returns the partial
    Gaussian product, which nothing consumes before the fold completes)."""
    prior: object
    folded: int


def _is_pointmass(q):
    return isinstance(q, PointMass)


def _mean_w(q_w):
    return float(q_w.mean()) if hasattr(q_w, "mean") else float(q_w)


def _elog_w(q_w):
    if isinstance(q_w, GammaShapeRate):
        return q_w.mean_log()
    w = _mean_w(q_w)
    return math.log(w) if w > 0.0 else float("nan")      # only the average energy uses it


def _engine(meta: UniSGPMeta, n: int):
    """The device object behind a meta; created on first use (raises without the HIP library / a GPU)."""
    M, D = meta.Xu.shape
    eng = meta.engine
    if eng is None or getattr(eng, "n_max", 0) < n:
        from .device import SGPDevice
        if eng is not None:
            eng.close()
        eng = SGPDevice(max(n, meta.N, 1), M, D, 1, device=meta.device, keep_kuf=True)
        meta.engine = eng
        meta._batch.pop("inducing_set", None)
    if not meta._batch.get("inducing_set"):
        eng.set_inducing(meta.Xu)
        meta._batch["inducing_set"] = True
    return eng


# ------------------------------------------------------------------------------------------------
# :v  (GPnode/UniSGPnode.jl:144-158 regression, :161-173 classification)
# ------------------------------------------------------------------------------------------------
def rule_v(q_out, q_in, q_w, q_theta, meta: UniSGPMeta) -> BufferUniSGP:
    if not _is_pointmass(q_theta):
        raise TypeError("q_theta must be a PointMass")
    mu_y = float(q_out.mean())
    v_y = 0.0 if _is_pointmass(q_out) else float(q_out.var())
    if not _is_pointmass(q_in):
        # uncertain input (GPnode/UniSGPnode.jl:125-140): the Psi-statistics are cubature sums, i.e. the same
        # statistics over the cubature points with their weights; the rule also adds 1e-8 I to every Psi2 (:135)
        if meta.method is None:
            raise ValueError("UniSGP(:v) with an uncertain input needs meta.method (a cubature rule)")
        pts, wts = meta.method.points_weights(q_in.mean(), q_in.var())
        x = (np.asarray(pts, dtype=np.float64), np.asarray(wts, dtype=np.float64))
        meta._batch["uncertain"] = True
    else:
        x = np.atleast_1d(np.asarray(q_in.mean(), dtype=np.float64))
        if meta._pending and meta._batch.get("uncertain"):
            raise NotImplementedError("PointMass and uncertain inputs mixed in one graph")
    w = _mean_w(q_w)
    theta = np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64))
    if meta._pending and (meta._batch.get("w") != w or not np.array_equal(meta._batch.get("theta"), theta)):
        raise ValueError("all UniSGP nodes of one graph share q_w and q_theta")
    if not meta._pending:
        meta._batch.update(w=w, theta=theta, E_logw=_elog_w(q_w), classification=False,
                           uncertain=not _is_pointmass(q_in))
    if not _is_pointmass(q_out):
        meta._batch["classification"] = True
    meta._pending.append((x, mu_y, v_y))
    return BufferUniSGP(len(meta._pending) - 1, meta)


# ------------------------------------------------------------------------------------------------
# prod(GenericProd, left::Normal, right::BufferUniSGP)   (GPnode/UniSGPnode.jl:62-73)
# ------------------------------------------------------------------------------------------------
def prod(left, right: BufferUniSGP):
    meta = right.meta
    if isinstance(left, PendingMarginal):
        prior = left.prior
    else:
        prior = left
    meta.counter += 1
    if meta.counter == 1:
        meta._prior = prior
    if meta.counter < meta.N:
        return PendingMarginal(meta._prior, meta.counter)
    if len(meta._pending) != meta.N:
        raise RuntimeError(f"meta.N = {meta.N} but {len(meta._pending)} UniSGP(:v) messages were produced; meta.N must "
                           "equal the number of UniSGP nodes in the graph (experiments/regression_kin40k.ipynb:155)")
    # ---- the N-th message: one device sweep for the whole batch
    uncertain = bool(meta._batch.get("uncertain"))
    extra_precision = 0.0
    if uncertain:
        X = np.concatenate([p[0][0] for p in meta._pending])
        wts = np.concatenate([p[0][1] for p in meta._pending])
        y = np.concatenate([np.full(len(p[0][1]), p[1]) for p in meta._pending])
        vy = None
        extra_precision = 1e-8 * meta._batch["w"] * meta.N             # N messages, each with Psi2 + 1e-8 I (:135,138)
    else:
        X = np.stack([p[0] for p in meta._pending])
        y = np.array([p[1] for p in meta._pending])
        vy = np.array([p[2] for p in meta._pending]) if meta._batch["classification"] else None
        wts = None
    eng = _engine(meta, len(y))
    sigma2, ell = meta.kernel(meta._batch["theta"])
    eng.set_data(X, y, vy, wts, n_nodes=meta.N)
    eng.set_kernel(sigma2, ell, meta.jitter)
    eng.set_noise([[meta._batch["w"]]], meta._batch["E_logw"])
    prior = meta._prior
    if extra_precision:
        from .device import potri
        if isinstance(prior, MvNormalMeanCovariance):
            L0 = potri(prior.S, meta.device)
            prior = MvNormalWeightedMeanPrecision(L0 @ prior.m, L0)
        elif isinstance(prior, MvNormalMeanPrecision):
            prior = MvNormalWeightedMeanPrecision(prior.W @ prior.m, prior.W)
        prior = MvNormalWeightedMeanPrecision(prior.xi, prior.W + extra_precision * np.eye(len(prior.xi)))
    if isinstance(prior, MvNormalMeanCovariance):
        eng.set_prior_meancov(prior.m, prior.S)
    elif isinstance(prior, MvNormalWeightedMeanPrecision):
        eng.set_prior_precision(prior.xi, prior.W)
    elif isinstance(prior, MvNormalMeanPrecision):
        eng.set_prior_precision(prior.W @ prior.m, prior.W)
    else:
        raise TypeError(f"unsupported prior type {type(prior).__name__}")
    eng.sweep()
    mu_v, Sigma_v, Uv = eng.posterior()            # raises PosDefException like cholesky() in the reference
    meta.Uv = Uv                                   # :69
    meta.KuuL = eng.kuu_chol()
    meta.counter = 0                               # :70
    index = {} if uncertain else {x.tobytes(): i for i, x in enumerate(X)}
    meta._batch.update(X=X, y=y, vy=vy, index=index, I=None, mu_v=mu_v)
    meta._pending = []
    return MvNormalMeanCovariance(mu_v, Sigma_v)


def _point_stats(q_in, q_v, meta: UniSGPMeta):
    """(I1_n, I2_n) of one point of the last swept batch, from the device's per-point pass."""
    if not _is_pointmass(q_in):
        raise NotImplementedError("uncertain inputs are not on the device path yet")
    if "index" not in meta._batch:
        raise RuntimeError("UniSGP(:w)/average energy before q(v) was computed for this batch")
    if not np.allclose(np.asarray(q_v.mean()), meta._batch["mu_v"], rtol=1e-12, atol=0):
        raise NotImplementedError("q_v differs from the marginal of the last sweep: the device evaluates the :w rule "
                                  "at its resident posterior")
    x = np.atleast_1d(np.asarray(q_in.mean(), dtype=np.float64))
    try:
        i = meta._batch["index"][x.tobytes()]
    except KeyError as e:
        raise KeyError("this input point was not part of the last swept batch") from e
    if meta._batch["I"] is None:
        meta._batch["I"] = meta.engine.w_stats()
    I1, I2 = meta._batch["I"]
    return float(I1[i]), float(I2[i])


# ------------------------------------------------------------------------------------------------
# :w  (GPnode/UniSGPnode.jl:196-216 regression, :219-238 classification)
# ------------------------------------------------------------------------------------------------
def rule_w(q_out, q_in, q_v, q_theta, meta: UniSGPMeta) -> GammaShapeRate:
    I1, I2 = _point_stats(q_in, q_v, meta)
    return GammaShapeRate(1.5, 0.5 * (I1 + I2))


def rule_w_summed(meta: UniSGPMeta, prior: GammaShapeRate) -> GammaShapeRate:
    """q(w) = prior x all N messages, from the sweep's reduced statistics (no per-point pass):
    shape a0 + N/2, rate b0 + (sum I1 + sum I2)/2."""
    sc = meta.engine.scalars()
    return GammaShapeRate(prior.a + 0.5 * meta.N, prior.b + 0.5 * (sc.sum_I1 + sc.sum_I2))


# ------------------------------------------------------------------------------------------------
# :out  (GPnode/UniSGPnode.jl:96-104)
# ------------------------------------------------------------------------------------------------
def rule_out(q_in, q_v, q_w, q_theta, meta: UniSGPMeta) -> NormalMeanPrecision:
    if not _is_pointmass(q_in):
        # GPnode/UniSGPnode.jl:85-93: Psi1 = sum_s omega_s K(Xu, x_s), mean = Psi1 . mu_v
        pts, wts = meta.method.points_weights(q_in.mean(), q_in.var())
        f = predict(np.asarray(pts, dtype=np.float64), q_v, q_theta, meta)
        return NormalMeanPrecision(float(np.dot(wts, f)), _mean_w(q_w))
    m = predict(np.atleast_2d(np.asarray(q_in.mean(), dtype=np.float64)), q_v, q_theta, meta)[0]
    return NormalMeanPrecision(float(m), _mean_w(q_w))


def predict(Xstar, q_v, q_theta, meta: UniSGPMeta) -> np.ndarray:
    """Batched `@call_rule UniSGP(:out)` (the 30 000-call loop of experiments/regression_kin40k.ipynb:288-304)."""
    Xstar = np.asarray(Xstar, dtype=np.float64)
    if Xstar.ndim == 1:
        Xstar = Xstar[:, None]
    eng = _engine(meta, 1)
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(q_theta.mean(), dtype=np.float64)))
    eng.set_kernel(sigma2, ell, meta.jitter)
    return eng.predict(Xstar, np.asarray(q_v.mean(), dtype=np.float64))


# ------------------------------------------------------------------------------------------------
# @average_energy  (GPnode/UniSGPnode.jl:337-359 Gamma w; :363-387 classification; :411-436 PointMass w)
# ------------------------------------------------------------------------------------------------
def average_energy(q_out, q_in, q_v, q_w, q_theta, meta: UniSGPMeta) -> float:
    I1, I2 = _point_stats(q_in, q_v, meta)
    w_bar = _mean_w(q_w)
    return 0.5 * (I1 * w_bar - _elog_w(q_w) + LOG2PI + I2 * w_bar)


def average_energy_summed(meta: UniSGPMeta) -> float:
    """Sum over the N nodes, from the sweep's scalars (w and E[log w] as given to the sweep)."""
    return meta.engine.scalars().energy


# ------------------------------------------------------------------------------------------------
# cold rules: present in the interface, not on the device path this round
# ------------------------------------------------------------------------------------------------
def rule_in(q_out, q_v, q_w, q_theta, meta: UniSGPMeta):
    raise NotImplementedError("UniSGP(:in) (GPnode/UniSGPnode.jl:107-122) returns a log-pdf closure evaluated by "
                              "quadrature in ReactiveMP; not on the device path yet (SURVEY.md §8 a15)")


def rule_theta(q_out, q_in, q_v, q_w, meta: UniSGPMeta):
    raise NotImplementedError("UniSGP(:theta) (GPnode/UniSGPnode.jl:242-287) returns a log-pdf closure; the "
                              "hyper-parameter objective is SURVEY.md §8 f1 (sgp_theta_objective)")
