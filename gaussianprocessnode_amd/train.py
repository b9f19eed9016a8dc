"""Streaming minibatch driver: the loop of `PerformInference` / `my_free_energy`
(experiments/regression_kin40k.ipynb:182-230) on the device path.

Per epoch the prior is reset to N(0, prior_var I) (:203-204); each minibatch runs one VMP sweep with the previous
minibatch's posterior as prior (:205-212), then one optimiser step on the kernel hyper-parameters with q(v) held fixed
(:213-222, `grad_llh_new!` + `Flux.Optimise.update!`).  Host logic only -- the numbers come from the engine (C ABI)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .distributions import MvNormalMeanCovariance
from .meta import softplus, split2batch


@dataclass
class AdaMax:
    """Flux.AdaMax (eta = 0.001, beta = (0.9, 0.999), eps = 1e-8) -- the optimiser of the kin40k / banana notebooks."""
    eta: float = 1e-3
    beta: tuple = (0.9, 0.999)
    eps: float = 1e-8
    _state: dict = field(default_factory=dict, repr=False)

    def update(self, x: np.ndarray, grad: np.ndarray) -> np.ndarray:
        st = self._state.setdefault(id(x), dict(m=np.zeros_like(x), u=np.zeros_like(x), bp=np.array(self.beta, dtype=float)))
        st["m"] = self.beta[0] * st["m"] + (1.0 - self.beta[0]) * grad
        st["u"] = np.maximum(self.beta[1] * st["u"], np.abs(grad))
        delta = (self.eta / (1.0 - st["bp"][0])) * st["m"] / (st["u"] + self.eps)
        st["bp"] = st["bp"] * np.array(self.beta)
        x -= delta
        return x


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))


def perform_inference(theta, xtrain, ytrain, Xu, engine, *, batch_size=500, epochs=1, w_val=1e4, prior_var=50.0,
                      jitter=0.0, optimizer=None, learn_theta=True, device_paced=None):
    """Returns (q_v of the last minibatch, theta) like `PerformInference` (:196-230).  `theta` is the raw
    (pre-softplus) parameter vector of `kernel_gp` (:108); `engine` an SGPDevice sized for `batch_size` points.

    device_paced (default: whenever the engine offers `train_begin` and the optimizer is fresh): the training set, theta and
    the optimiser state stay on the device and the loop below only enqueues (sgp_train_* in include/sgp_hip.h); otherwise
    every minibatch goes through the setters, `theta_objective` and the host-side `AdaMax` (the same arithmetic,
    host-paced).  Device-paced, a minibatch whose K_uu or Lambda is not positive definite is skipped and counted; the
    LinAlgError is raised after the run, not at the failing minibatch."""
    theta = np.array(theta, dtype=np.float64)
    xtrain = np.asarray(xtrain, dtype=np.float64).reshape(len(ytrain), -1)
    Xu = np.asarray(Xu, dtype=np.float64).reshape(-1, xtrain.shape[1])
    M = Xu.shape[0]
    optimizer = optimizer or AdaMax()
    if device_paced is None:
        # (the device-paced run starts AdaMax from zero state and does not write the moments back: an optimizer that has
        # already taken steps keeps to the host-paced loop, which continues where it left off)
        device_paced = hasattr(engine, "train_begin") and not optimizer._state
    if device_paced:
        return _perform_inference_device(theta, xtrain, ytrain, Xu, engine, batch_size, epochs, w_val, prior_var, jitter,
                                         optimizer, learn_theta)
    xb, yb = split2batch((xtrain, np.asarray(ytrain, dtype=np.float64)), batch_size)
    engine.set_inducing(Xu)
    engine.set_noise([[w_val]])
    device_carry = hasattr(engine, "carry_posterior")
    mu, Sigma = np.zeros(M), prior_var * np.eye(M)
    zero, Lam_prior = np.zeros(M), np.eye(M) / prior_var
    for _ in range(epochs):
        mu, Sigma = np.zeros(M), prior_var * np.eye(M)                     # :203-204
        if device_carry:
            engine.set_prior_precision(zero, Lam_prior)                    # dense form: the captured graphs stay valid
        for xi, yi in zip(xb, yb):
            p = softplus(theta)
            if not device_carry:
                engine.set_prior_meancov(mu, Sigma)
            engine.set_data(xi, yi)
            engine.set_kernel(float(p[0]), p[1:], jitter)                  # :183-184 (no jitter in training)
            engine.sweep()                                                 # :185-192  infer(iterations = 1)
            if device_carry:
                engine.carry_posterior()                                   # :212 without leaving the device
            else:
                mu, Sigma, _ = engine.posterior(want_uv=False)
            if learn_theta:
                _, g = engine.theta_objective(want_grad=True, n_ell=len(p) - 1)    # :214-221
                optimizer.update(theta, g * sigmoid(theta))                # chain rule through softplus; :222
    if device_carry:
        mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), theta


def _perform_inference_device(theta, xtrain, ytrain, Xu, engine, batch_size, epochs, w_val, prior_var, jitter, optimizer,
                              learn_theta):
    """The same loop with the device pacing itself: one `train_step` per minibatch, an isotropic-prior reset per epoch
    (:203-204); the host waits once, at the end."""
    N = len(ytrain)
    if optimizer._state:
        raise ValueError("perform_inference(device_paced=True) starts AdaMax from zero state; pass a fresh optimizer")
    engine.set_inducing(Xu)
    engine.set_noise([[w_val]])
    engine.set_prior_isotropic(prior_var)                                  # :203-204, put back by reset_prior every epoch
    engine.train_begin(xtrain, ytrain, theta, jitter=jitter, eta=optimizer.eta, beta=optimizer.beta, eps=optimizer.eps)
    for _ in range(epochs):
        for o in range(0, N, batch_size):
            engine.train_step(o, min(batch_size, N - o), learn_theta, reset_prior=(o == 0))
    theta, _, skipped = engine.train_end()
    if skipped:
        raise np.linalg.LinAlgError(f"{skipped} minibatch(es) had a K_uu or Lambda that is not positive definite")
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), np.asarray(theta)


def probit_marginal(y, mz, vz):
    """q(f) for `y ~ Probit(f)` with the forward message N(f; mz, vz): the moment-matched product of the UniSGP :out
    message with ReactiveMP's Probit(:in) message for a PointMass output (y in {0, 1}).  Returns (mean, variance)."""
    from scipy.special import log_ndtr
    y = np.asarray(y, dtype=np.float64)
    mz = np.asarray(mz, dtype=np.float64)
    s = 2.0 * y - 1.0
    g = s * mz / np.sqrt(1.0 + vz)
    r = np.exp(-0.5 * g * g - 0.5 * np.log(2.0 * np.pi) - log_ndtr(g))        # phi(g) / Phi(g)
    return mz + s * vz * r / np.sqrt(1.0 + vz), vz - vz * vz / (1.0 + vz) * r * (g + r)


def vmp_regression(p, xtrain, ytrain, Xu, engine, *, iterations=7, prior_var=50.0, shape=1e-2, rate=1e-2, jitter=1e-8):
    """The inner `infer(...)` of experiments/GPT_regression.ipynb's `my_free_energy` (cell 9; model cell 6:
    `v ~ MvNormal(0, 50 I); w ~ Gamma(1e-2, 1e-2); y[i] ~ UniSGP(x[i], v, w, theta)`, mean field q(v) q(w), q(w) initialised
    at its prior, 7 iterations) at the kernel parameters p = (sigma2, lengthscale...) -- BASELINE config 1.  Per iteration:
    one sweep for q(v) at mean(q_w), then q(w) = Gamma(shape + N/2, rate + (sum I1 + sum I2)/2) from that q(v)
    (GPnode/UniSGPnode.jl:56-73,196-238).  Returns (q_v, (shape, rate))."""
    xtrain = np.asarray(xtrain, dtype=np.float64).reshape(len(ytrain), -1)
    Xu = np.asarray(Xu, dtype=np.float64).reshape(-1, xtrain.shape[1])
    p = np.asarray(p, dtype=np.float64)
    a, b = float(shape), float(rate)
    engine.set_inducing(Xu)
    engine.set_kernel(float(p[0]), p[1:], jitter)
    engine.set_prior_isotropic(prior_var)
    engine.set_data(xtrain, np.asarray(ytrain, dtype=np.float64))
    for _ in range(iterations):
        engine.set_noise([[a / b]])
        engine.sweep()
        sc = engine.scalars()
        a, b = shape + 0.5 * len(ytrain), rate + 0.5 * (sc.sum_I1 + sc.sum_I2)
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), (a, b)


def vmp_classification(p, xtrain, ytrain, Xu, engine, *, iterations=30, prior_var=50.0, shape=1e-2, rate=1e-2, jitter=0.0):
    """The inner `infer(...)` of experiments/GPT_classification.ipynb's `my_free_energy` (cell 9; model cell 7:
    `f[i] ~ UniSGP(x[i], v, w, theta); y[i] ~ Probit(f[i])`, mean field q(f) q(v) q(w), q(v) and q(w) initialised at their
    priors, 30 iterations, K_uu without jitter).  Per iteration: q(f_i) from the :out message N(k_i' mu_v, 1 / mean(q_w))
    and the Probit likelihood, one sweep for q(v) with q_out = q(f), then q(w).  Returns (q_v, (shape, rate))."""
    xtrain = np.asarray(xtrain, dtype=np.float64).reshape(len(ytrain), -1)
    ytrain = np.asarray(ytrain, dtype=np.float64)
    Xu = np.asarray(Xu, dtype=np.float64).reshape(-1, xtrain.shape[1])
    p = np.asarray(p, dtype=np.float64)
    a, b = float(shape), float(rate)
    mu = np.zeros(Xu.shape[0])
    engine.set_inducing(Xu)
    engine.set_kernel(float(p[0]), p[1:], jitter)
    engine.set_prior_isotropic(prior_var)
    for _ in range(iterations):
        w = a / b
        mf, vf = probit_marginal(ytrain, engine.predict(xtrain, mu), 1.0 / w)
        engine.set_data(xtrain, mf, vf)
        engine.set_noise([[w]])
        engine.sweep()
        sc = engine.scalars()
        a, b = shape + 0.5 * len(ytrain), rate + 0.5 * (sc.sum_I1 + sc.sum_I2)
        mu = engine.posterior(want_cov=False, want_uv=False)[0]
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), (a, b)


def perform_inference_classification(theta, xtrain, ytrain, Xu, engine, *, batch_size=200, epochs=1, prior_var=50.0,
                                     shape=0.01, rate=0.01, jitter=1e-8, optimizer=None, device_paced=None):
    """`PerformInference` of experiments/classification_banana.ipynb (model `f[i] ~ UniSGP(x[i], v, w, theta);
    y[i] ~ Probit(f[i])`, mean-field q(f) q(v) q(w), one VMP iteration per minibatch, q(v) and q(w) carried over every
    minibatch and never reset).  Per minibatch:
      q(f_i)  from the :out message N(k_i mu_v, 1 / mean(q_w)) (GPnode/UniSGPnode.jl:96-104) and the Probit likelihood;
      q(v)    one sweep with q_out = q(f_i) (the classification :v rule, :161-173);
      q(w)    Gamma(a + n/2, b + (sum I1 + sum I2)/2) with the NEW q(v) and the `meta.Uv` its product hook just stored
              (:56-73, :219-238);
      theta   one optimiser step on neg_log_backwardmess_fast with y_data = mean(q_f), w = mean(new q_w).
    Returns (q_v, (shape, rate), theta).

    The order of the updates inside the single VMP iteration is the reference scheduler's business (RxInfer / ReactiveMP, no
    pinned version); this is the order in which `meta.Uv` is refreshed by the product hook before the :w messages read it.
    Other orders were measured (tests/scripts/banana_schedules.py, profiles/*_train_banana_schedules.jsonl): none reproduces the
    reference's saved end point, see DESIGN.md section 2.

    device_paced (default: whenever the engine offers `train_begin`): the whole loop -- Probit moment matching, the Gamma
    update and AdaMax included -- runs on the device (sgp_train_begin with SGP_LIKELIHOOD_PROBIT); the host only enqueues."""
    theta = np.array(theta, dtype=np.float64)
    xtrain = np.asarray(xtrain, dtype=np.float64).reshape(len(ytrain), -1)
    ytrain = np.asarray(ytrain, dtype=np.float64)
    Xu = np.asarray(Xu, dtype=np.float64).reshape(-1, xtrain.shape[1])
    M = Xu.shape[0]
    optimizer = optimizer or AdaMax()
    if device_paced is None:
        device_paced = hasattr(engine, "train_begin") and not optimizer._state
    if device_paced:
        return _perform_inference_classification_device(theta, xtrain, ytrain, Xu, engine, batch_size, epochs, prior_var, shape,
                                                        rate, jitter, optimizer)
    xb, yb = split2batch((xtrain, ytrain), batch_size)
    a, b = float(shape), float(rate)
    engine.set_inducing(Xu)
    engine.set_prior_precision(np.zeros(M), np.eye(M) / prior_var)
    mu = np.zeros(M)
    first = True
    for _ in range(epochs):
        for xi, yi in zip(xb, yb):
            p = softplus(theta)
            w0 = a / b
            engine.set_kernel(float(p[0]), p[1:], jitter)
            mz = engine.predict(xi, mu if first else None)             # k_i' mu_v with the carried posterior mean
            mf, vf = probit_marginal(yi, mz, 1.0 / w0)
            engine.set_data(xi, mf, vf)
            engine.set_noise([[w0]])
            engine.sweep()
            sc = engine.scalars()
            a, b = a + 0.5 * len(yi), b + 0.5 * (sc.sum_I1 + sc.sum_I2)
            engine.carry_posterior()
            engine.set_noise([[a / b]])                                # grad_llh_new!(...; w = mean(qw))
            _, g = engine.theta_objective(want_grad=True, n_ell=len(p) - 1)
            optimizer.update(theta, g * sigmoid(theta))
            first = False
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), (a, b), theta


def _perform_inference_classification_device(theta, xtrain, ytrain, Xu, engine, batch_size, epochs, prior_var, shape, rate, jitter,
                                             optimizer):
    """The same loop paced by the device: one `train_step` per minibatch (window of the resident set; the forward message, the
    Probit moments, the Gamma update and AdaMax are kernels between the sweep's own), the host waits once, at the end."""
    N = len(ytrain)
    engine.set_inducing(Xu)
    engine.set_prior_isotropic(prior_var)                                  # q(v) starts at its prior and is never reset
    engine.train_begin(xtrain, ytrain, theta, jitter=jitter, eta=optimizer.eta, beta=optimizer.beta, eps=optimizer.eps,
                       likelihood="probit", gamma=(shape, rate))
    for _ in range(epochs):
        for o in range(0, N, batch_size):
            engine.train_step(o, min(batch_size, N - o), True, reset_prior=False)
    theta, _, skipped = engine.train_end()
    if skipped:
        raise np.linalg.LinAlgError(f"{skipped} minibatch(es) had a K_uu or Lambda that is not positive definite")
    a, b = engine.train_gamma()
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), (a, b), np.asarray(theta)
