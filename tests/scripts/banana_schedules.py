#!/usr/bin/env python3
"""Experiment scaffolding that used to live in gaussianprocessnode_amd/train.py: the banana classification loop
(experiments/classification_banana.ipynb cell 9) with the ORDER of the updates inside the single VMP iteration as a knob, plus
a gradient-only jitter and a per-epoch reset of q(v).  None of the orders reproduces the reference's saved end point
(softplus(theta) = [0.986, 1.028, 1.022], q(w) rate 1.72e6); results are appended to profiles/*_train_banana_schedules.jsonl.

    python tests/scripts/banana_schedules.py --w-schedule new_mu_old_uv [--epochs 500]

Schedules: after_v (the product driver's), before_v, w_then_v, f_again, f_again_w (round 2), and the two mixed ones of round 3 --
the reference's :w rule takes mu_v from the q_v MARGINAL and Uv from META, two separately updated states, and the notebook seeds
meta.Uv with the prior's factor (`Lu = fastcholesky!(mv mv' + Sigma_v).U`, cell 9) because :w may fire before the product hook:
  new_mu_old_uv   q(w) from the new q(v)'s mean and the Uv the iteration started with (the seed `Lu`)
  old_mu_new_uv   q(w) from the mean the iteration started with and the Uv the product hook just stored
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussianprocessnode_amd.distributions import MvNormalMeanCovariance
from gaussianprocessnode_amd.meta import softplus, split2batch
from gaussianprocessnode_amd.train import AdaMax, probit_marginal, sigmoid

SCHEDULES = ["after_v", "before_v", "w_then_v", "f_again", "f_again_w", "new_mu_old_uv", "old_mu_new_uv"]


def run_schedule(theta, xtrain, ytrain, Xu, engine, *, batch_size=200, epochs=1, prior_var=50.0,
                                     shape=0.01, rate=0.01, jitter=1e-8, optimizer=None, w_schedule="after_v",
                                     grad_jitter=None, reset_v_each_epoch=False):
    """`PerformInference` of experiments/classification_banana.ipynb (model `f[i] ~ UniSGP(x[i], v, w, theta);
    y[i] ~ Probit(f[i])`, mean-field q(f) q(v) q(w), one VMP iteration per minibatch, q(v) and q(w) carried over every
    minibatch and never reset).  Per minibatch:
      q(f_i)  from the :out message N(k_i mu_v, 1 / mean(q_w)) (GPnode/UniSGPnode.jl:96-104) and the Probit likelihood;
      q(v)    one sweep with q_out = q(f_i) (the classification :v rule, :161-173);
      q(w)    Gamma(a + n/2, b + (sum I1 + sum I2)/2) with the NEW q(v) and the `meta.Uv` its product hook just stored
              (:56-73, :219-238);
      theta   one optimiser step on neg_log_backwardmess_fast with y_data = mean(q_f), w = mean(new q_w).
    Returns (q_v, (shape, rate), theta).

    w_schedule: the order of the updates inside the single VMP iteration -- "after_v" (default: q(w) from the minibatch's
    new q(v) and the `meta.Uv` its product hook just stored), "before_v" (from the q(v) the iteration started with),
    "w_then_v" (as before_v, and the sweep already uses the new mean(q_w)), "f_again_w" (q(f) recomputed from the new q(v)
    before q(w)), "f_again" (only the gradient sees the recomputed q(f)).  None reproduces the reference's end point
    (softplus(theta) = [0.986, 1.028, 1.022], q(w) rate 1.72e6): rates 5.9e5 / 3.6e9 / 2.5e9 / 7.2e5 / 5.9e5
    (profiles/r02_train_banana_schedules.jsonl).  mean(q_w) is neutrally stable (b/a = mean(I1 + I2) ~ 1/mean(q_w) holds
    for any value), so it follows the update order; the K_uu treatment of the reference's gradient (no jitter,
    derivative_helper.jl:24-25) is not the lever: tools/banana_gradient_probe.py bounds its effect on the first gradient at
    7e-5 .. 6e-3, and `grad_jitter` (a different jitter in the gradient only) from 1e-11 to 1e-4 leaves the end point where
    it is.  `reset_v_each_epoch` (the notebook's commented-out lines) does not reproduce it either.  DESIGN.md section 2."""
    theta = np.array(theta, dtype=np.float64)
    xtrain = np.asarray(xtrain, dtype=np.float64).reshape(len(ytrain), -1)
    ytrain = np.asarray(ytrain, dtype=np.float64)
    Xu = np.asarray(Xu, dtype=np.float64).reshape(-1, xtrain.shape[1])
    M = Xu.shape[0]
    optimizer = optimizer or AdaMax()
    xb, yb = split2batch((xtrain, ytrain), batch_size)
    a, b = float(shape), float(rate)
    engine.set_inducing(Xu)
    engine.set_prior_precision(np.zeros(M), np.eye(M) / prior_var)
    mu = np.zeros(M)
    Uv_old = np.sqrt(prior_var) * np.eye(M)
    first = True
    for _ in range(epochs):
        if reset_v_each_epoch and not first:                               # (experiment: q(v) back to its prior every epoch, q(w) kept)
            engine.set_prior_precision(np.zeros(M), np.eye(M) / prior_var)
            mu = np.zeros(M)
            Uv_old = np.sqrt(prior_var) * np.eye(M)
            first = True
        for xi, yi in zip(xb, yb):
            p = softplus(theta)
            w0 = a / b
            engine.set_kernel(float(p[0]), p[1:], jitter)
            mz = engine.predict(xi, mu if first else None)             # k_i' mu_v with the carried posterior mean
            mf, vf = probit_marginal(yi, mz, 1.0 / w0)
            engine.set_data(xi, mf, vf)
            engine.set_noise([[w0]])
            carried = False
            if w_schedule in ("before_v", "w_then_v"):
                # q(w) from the q(v) this iteration STARTED with: the per-point I1 / I2 at the carried posterior, before the
                # sweep replaces it ("w_then_v": the sweep then already uses the new mean(q_w))
                engine.sweep_local()
                engine.set_posterior(mu, Uv_old)
                I1, I2 = engine.w_stats()
                s_I = float(np.sum(I1) + np.sum(I2))
                if w_schedule == "w_then_v":
                    engine.set_noise([[(a + 0.5 * len(yi)) / (b + 0.5 * s_I)]])
            engine.sweep()
            if w_schedule in ("after_v", "f_again", "f_again_w"):
                sc = engine.scalars()
                s_I = sc.sum_I1 + sc.sum_I2
            elif w_schedule in ("new_mu_old_uv", "old_mu_new_uv"):
                # the :w rule with mu_v from one state and Uv from the other (marginal vs meta, see the module docstring)
                mu_n, _, Uv_n = engine.posterior(want_cov=False)
                engine.carry_posterior()                               # (now: set_posterior below detaches q(v) from the statistics)
                carried = True
                engine.set_posterior(mu_n if w_schedule == "new_mu_old_uv" else mu, Uv_old if w_schedule == "new_mu_old_uv" else Uv_n)
                I1, I2 = engine.w_stats()
                s_I = float(np.sum(I1) + np.sum(I2))
                engine.set_posterior(mu_n, Uv_n)                       # (the carry and the gradient see the sweep's q(v))
                mu, Uv_old = mu_n, Uv_n
            else:
                mu, _, Uv_old = engine.posterior(want_cov=False)
            if w_schedule == "f_again_w":
                # q(f) once more from the NEW q(v) (still at the old mean(q_w)), and q(w) from that q(f) and the new q(v)
                mu_n, _, Uv_n = engine.posterior(want_cov=False)
                mf, vf = probit_marginal(yi, engine.predict(xi, None), 1.0 / w0)
                engine.carry_posterior()
                engine.set_data(xi, mf, vf)
                engine.sweep_local()
                engine.set_posterior(mu_n, Uv_n)
                I1, I2 = engine.w_stats()
                s_I = float(np.sum(I1) + np.sum(I2))
                a, b = a + 0.5 * len(yi), b + 0.5 * s_I
            else:
                a, b = a + 0.5 * len(yi), b + 0.5 * s_I
                if not carried:
                    engine.carry_posterior()
                if w_schedule == "f_again":
                    # the q(f) the iteration ENDS with: recomputed from the new q(v) and the new mean(q_w); it is what the
                    # gradient then sees as y_data (the statistics are re-formed with it, q(v) re-installed unchanged)
                    mu_n, _, Uv_n = engine.posterior(want_cov=False)
                    mf, vf = probit_marginal(yi, engine.predict(xi, None), b / a)
                    engine.set_data(xi, mf, vf)
                    engine.sweep_local()
                    engine.set_posterior(mu_n, Uv_n)
            engine.set_noise([[a / b]])                                # grad_llh_new!(...; w = mean(qw))
            if grad_jitter is not None:                                # (experiment: a different K_uu jitter in the gradient only)
                engine.set_kernel(float(p[0]), p[1:], grad_jitter)
            _, g = engine.theta_objective(want_grad=True, n_ell=len(p) - 1)
            optimizer.update(theta, g * sigmoid(theta))
            first = False
    mu, Sigma, _ = engine.posterior(want_uv=False)
    return MvNormalMeanCovariance(mu, Sigma), (a, b), theta


def main():
    import gaussianprocessnode_amd as G
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--w-schedule", choices=SCHEDULES, default="after_v")
    ap.add_argument("--grad-jitter", type=float, default=None)
    ap.add_argument("--reset-v", action="store_true")
    args = ap.parse_args()
    fix = np.load(os.path.join(ROOT, "tests", "golden", "banana_fixture.npz"))
    data = fix["data"]
    X, lab = data[:, :2], np.where(data[:, 2] < 0, 0.0, data[:, 2])
    xtrain, ytrain, xtest, ytest = X[:4000], lab[:4000], X[4000:], lab[4000:]
    Xu = fix["Xu"]
    M, D = Xu.shape
    theta_init = np.log(np.expm1(np.ones(D + 1)))
    with G.SGPDevice(args.batch, M, D, keep_kuf=(args.w_schedule != "after_v")) as dev:
        t0 = time.perf_counter()
        qv, (a, b), theta = run_schedule(theta_init, xtrain, ytrain, Xu, dev, batch_size=args.batch, epochs=args.epochs,
                                         w_schedule=args.w_schedule, grad_jitter=args.grad_jitter, reset_v_each_epoch=args.reset_v)
        t_train = time.perf_counter() - t0
        p = softplus(theta)
        dev.set_kernel(float(p[0]), p[1:], 1e-8)
        pred = dev.predict(xtest, qv.m)
    errors = float(np.sum(np.abs((pred >= 0).astype(float) - ytest)))
    print(json.dumps({"w_schedule": args.w_schedule, "epochs": args.epochs, "grad_jitter": args.grad_jitter, "reset_v_each_epoch": args.reset_v,
                      "train_seconds": t_train, "errors": errors, "theta_softplus": [float(v) for v in p], "qw": [a, b],
                      "reference": {"errors": 125.0, "theta_softplus": [float(v) for v in softplus(fix["theta_opt"])],
                                    "qw": [float(v) for v in fix["qw_ab"]]}}))


if __name__ == "__main__":
    main()
