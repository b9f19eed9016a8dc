"""Host-side mirror of the `UniSGP` factor node (GPnode/UniSGPnode.jl): same rule names, argument meaning and
error behaviour, batched on the device.

The reference's rules run once per data point and the N-fold Gaussian product folds N rank-1 M x M messages
(GPnode/UniSGPnode.jl:62-73,144-173).  Here the per-point `:v` rule returns an O(1) token and the product runs
ONE device sweep when the N-th token is folded -- the same `counter == N` hook the reference uses to refresh
`meta.Uv` (:64-71).  The PointMass-input rules (`q_in::PointMass`) are the hot path; the uncertain-input variants run on
the device as cubature-weighted data: `:v` (:125-140) and `:out` (:85-93) inside the sweep, the per-node clamped `:w` /
average-energy variants (:177-192,290-313,390-409), the `:in` closure (:107-122) and the `:theta` closures (:242-287)
as stand-alone per-point evaluations at whatever q_v / meta.Uv the caller passes.

What runs where: Gram matrices, Psi-statistics, the factorisations, q(v), `meta.Uv`, the :w / energy sums and the
per-point quadratic forms come from `meta.engine` (`SGPDevice`, the C ABI); without the HIP library and a gfx950 GPU the
first sweep raises -- there is no CPU fallback.  What stays on the host is message bookkeeping: the token counter,
cubature points and weights of an uncertain input (cubature.py), scalar arithmetic on the returned sums (Gamma shape and
rate, the energy's constants), and the conversion of a caller-supplied q_v into the (mean, upper factor) pair the cold
rules upload -- that factor is taken on the device too (`potrf`).
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


def _aux_engine(meta: UniSGPMeta, n: int):
    """A second device object for stand-alone rule evaluations (it replaces data and posterior, so it must not be the
    engine that holds the last swept batch)."""
    M, D = meta.Xu.shape
    eng = meta._batch.get("aux_engine")
    if eng is None or eng.n_max < n:
        from .device import SGPDevice
        if eng is not None:
            eng.close()
        eng = SGPDevice(max(n, 64), M, D, 1, device=meta.device, keep_kuf=True)
        eng.set_inducing(meta.Xu)
        meta._batch["aux_engine"] = eng
    return eng


def _stats_at(meta: UniSGPMeta, theta, X, y, vy, mu_v, Uv):
    """Per-point (I1_n, I2_n) of GPnode/UniSGPnode.jl:196-238 for arbitrary points, at kernel(theta) and at the q(v) given by
    (mu_v, Uv = chol(Sigma_v + mu mu').U): K_uu chain and K_uf on the device, then the per-point quadratic forms."""
    X = np.asarray(X, dtype=np.float64).reshape(len(y), -1)
    eng = _aux_engine(meta, len(y))
    sigma2, ell = meta.kernel(np.atleast_1d(np.asarray(theta, dtype=np.float64)))
    eng.set_data(X, y, vy)
    eng.set_kernel(sigma2, ell, meta.jitter)
    eng.sweep_local()
    eng.set_posterior(mu_v, Uv)
    return eng.w_stats()


def _uv_of(q_v, meta: UniSGPMeta):
    """chol(Sigma_v + mu mu').U of an explicit q_v, factored on the device."""
    from .device import potrf
    mu, Sig = q_v.mean_cov()
    mu = np.asarray(mu, dtype=np.float64)
    return potrf(np.asarray(Sig, dtype=np.float64) + np.outer(mu, mu), meta.device).T


def _node_I(q_out, q_in, mu_v, Uv, theta, meta: UniSGPMeta, jitter_psi2: float, clamp: bool):
    """(I1, I2) of ONE node whose input is uncertain: Psi-statistics by meta.method's cubature, Psi2 + jitter_psi2 I, clamped
    like the reference (GPnode/UniSGPnode.jl:186-190)."""
    if meta.method is None:
        raise ValueError("an uncertain input needs meta.method (a cubature rule)")
    pts, wts = meta.method.points_weights(q_in.mean(), q_in.var())
    wts = np.asarray(wts, dtype=np.float64)
    mu_y = float(q_out.mean())
    v_y = 0.0 if _is_pointmass(q_out) else float(q_out.var())
    I1q, I2q = _stats_at(meta, theta, pts, np.full(len(wts), mu_y), None, mu_v, Uv)
    I1 = float(wts @ I1q)
    I2 = float(wts @ I2q) + mu_y * mu_y * (1.0 - float(wts.sum())) + v_y
    if jitter_psi2:
        from .device import potri
        KuuL = np.asarray(meta.KuuL, dtype=np.float64)
        I1 -= jitter_psi2 * float(np.trace(potri(KuuL @ KuuL.T, meta.device)))     # tr(Kuu^-1 (jitter I))
        I2 += jitter_psi2 * float(np.sum(np.asarray(Uv) ** 2))                      # tr(Uv'Uv (jitter I))
    if clamp:
        I1, I2 = min(max(I1, 1e-12), 1e12), min(max(I2, 1e-12), 1e12)
    return I1, I2


def _point_stats(q_out, q_in, q_v, q_theta, meta: UniSGPMeta):
    """(I1_n, I2_n) of one PointMass-input node: from the per-point pass over the last swept batch when the point and q_v
    belong to it, else evaluated stand-alone at (q_v, meta.Uv) like the reference's rule would."""
    x = np.atleast_1d(np.asarray(q_in.mean(), dtype=np.float64))
    mu_v = np.asarray(q_v.mean(), dtype=np.float64)
    b = meta._batch
    if "index" in b and x.tobytes() in b["index"] and np.allclose(mu_v, b["mu_v"], rtol=1e-12, atol=0):
        if b["I"] is None:
            b["I"] = meta.engine.w_stats()
        i = b["index"][x.tobytes()]
        return float(b["I"][0][i]), float(b["I"][1][i])
    v_y = None if _is_pointmass(q_out) else np.array([float(q_out.var())])
    I1, I2 = _stats_at(meta, q_theta.mean(), x[None, :], np.array([float(q_out.mean())]), v_y, mu_v, meta.Uv)
    return float(I1[0]), float(I2[0])


# ------------------------------------------------------------------------------------------------
# :w  (GPnode/UniSGPnode.jl:196-216 regression, :219-238 classification, :177-192 uncertain input)
# ------------------------------------------------------------------------------------------------
def rule_w(q_out, q_in, q_v, q_theta, meta: UniSGPMeta) -> GammaShapeRate:
    if _is_pointmass(q_in):
        I1, I2 = _point_stats(q_out, q_in, q_v, q_theta, meta)
    else:
        I1, I2 = _node_I(q_out, q_in, np.asarray(q_v.mean(), dtype=np.float64), meta.Uv, q_theta.mean(), meta, 1e-8, True)
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
    if _is_pointmass(q_in):
        I1, I2 = _point_stats(q_out, q_in, q_v, q_theta, meta)
    elif isinstance(q_w, GammaShapeRate):
        # :290-313 -- meta.KuuL / meta.Uv, Psi2 + 1e-8 I, clamped
        I1, I2 = _node_I(q_out, q_in, np.asarray(q_v.mean(), dtype=np.float64), meta.Uv, q_theta.mean(), meta, 1e-8, True)
    else:
        # :390-409 (q_w::PointMass) -- Sigma_v + mu mu' from q_v itself.  The reference also adds 1e-8 to EVERY ENTRY of
        # Kuu, Psi1 and Psi2 there (`.+ 1e-8`); that quirk moves the result by < 1e-6 (GPtest.jl:337-348 tests it against
        # the clean formula at 1e-6) and is not reproduced.
        I1, I2 = _node_I(q_out, q_in, np.asarray(q_v.mean(), dtype=np.float64), _uv_of(q_v, meta), q_theta.mean(), meta, 0.0, True)
    w_bar = _mean_w(q_w)
    return 0.5 * (I1 * w_bar - _elog_w(q_w) + LOG2PI + I2 * w_bar)


def average_energy_summed(meta: UniSGPMeta) -> float:
    """Sum over the N nodes, from the sweep's scalars (w and E[log w] as given to the sweep)."""
    return meta.engine.scalars().energy


# ------------------------------------------------------------------------------------------------
# :in and :theta return log-density closures (ContinuousUnivariateLogPdf / ContinuousMultivariateLogPdf in ReactiveMP);
# here a callable whose every evaluation is one batched device pass
# ------------------------------------------------------------------------------------------------
class LogPdfClosure:
    def __init__(self, fn, multivariate: bool = False):
        self.fn = fn
        self.multivariate = multivariate

    def logpdf(self, x):
        return self.fn(x)

    __call__ = logpdf


def rule_in(q_out, q_v, q_w, q_theta, meta: UniSGPMeta) -> LogPdfClosure:
    """GPnode/UniSGPnode.jl:107-122: x -> -w/2 A(x) + w mu_y B(x).mu_v - w/2 |Uv B(x)|^2 with A(x) = k(x,x) - |KuuL^-1 B(x)|^2
    = -w/2 (I1(x) + I2(x) - mu_y^2).  Accepts a scalar or an array of inputs (one device pass for all of them)."""
    w_bar, mu_y = _mean_w(q_w), float(q_out.mean())
    mu_v = np.asarray(q_v.mean(), dtype=np.float64)
    theta, Uv = q_theta.mean(), meta.Uv

    def log_backwardmess(x):
        xs = np.atleast_1d(np.asarray(x, dtype=np.float64))
        X = xs.reshape(-1, meta.Xu.shape[1])
        I1, I2 = _stats_at(meta, theta, X, np.full(len(X), mu_y), None, mu_v, Uv)
        val = -0.5 * w_bar * (I1 + I2 - mu_y * mu_y)
        return float(val[0]) if np.ndim(x) == 0 or (meta.Xu.shape[1] > 1 and np.ndim(x) == 1) else val
    return LogPdfClosure(log_backwardmess)


def rule_theta(q_out, q_in, q_v, q_w, meta: UniSGPMeta) -> LogPdfClosure:
    """GPnode/UniSGPnode.jl:242-287: theta -> w mu_y Psi1(theta).mu_v - w/2 (Psi0(theta) + tr(Psi2(theta) (Rv - Kuu^-1(theta))))
    = -w/2 (sum_q omega_q (I1_q + I2_q) - mu_y^2 sum_q omega_q), Rv from q_v itself.  The batch-summed version of this is
    the hyper-parameter objective `SGPDevice.theta_objective` (helper_functions/derivative_helper.jl:23-39)."""
    if not _is_pointmass(q_in) and meta.method is None:
        raise ValueError("UniSGP(:theta) with an uncertain input needs meta.method (a cubature rule)")
    w_bar, mu_y = _mean_w(q_w), float(q_out.mean())
    mu_v = np.asarray(q_v.mean(), dtype=np.float64)
    Uv = _uv_of(q_v, meta)
    if _is_pointmass(q_in):
        pts, wts = np.atleast_1d(np.asarray(q_in.mean(), dtype=np.float64))[None, :], np.ones(1)
    else:
        pts, wts = meta.method.points_weights(q_in.mean(), q_in.var())
    wts = np.asarray(wts, dtype=np.float64)

    def log_backwardmess(theta):
        I1, I2 = _stats_at(meta, theta, pts, np.full(len(wts), mu_y), None, mu_v, Uv)
        return float(-0.5 * w_bar * (wts @ (I1 + I2) - mu_y * mu_y * wts.sum()))
    return LogPdfClosure(log_backwardmess, multivariate=True)


def prod_logpdf(left, right):
    """ReactiveMP.prod(GenericProd, Gaussian, ContinuousUnivariateLogPdf) and its mirror image (GPnode/UniSGPnode.jl:39-54):
    moments of N(x) exp(logpdf(x)) by ghcubature(21); `v + 1e-6` only when the Gaussian is on the left; NaN moments return
    the Gaussian unchanged."""
    from .cubature import ghcubature
    gauss, closure, pad = (left, right, 1e-6) if isinstance(right, LogPdfClosure) else (right, left, 0.0)
    pts, wts = ghcubature(21).points_weights(gauss.mean(), gauss.var())
    g = np.exp(np.asarray(closure.logpdf(pts[:, 0]), dtype=np.float64))
    Z = float(wts @ g)
    m = float(wts @ (pts[:, 0] * g)) / Z if Z != 0.0 else float("nan")
    v = float(wts @ ((pts[:, 0] - m) ** 2 * g)) / Z if Z != 0.0 else float("nan")
    if math.isnan(m) or math.isnan(v):
        return gauss
    return NormalMeanVariance(m, v + pad)
