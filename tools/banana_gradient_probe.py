#!/usr/bin/env python3
"""Why the banana trajectory (BASELINE config 4, experiments/classification_banana.ipynb) cannot be pinned end to end.

The reference's theta gradient is ForwardDiff through `fastcholesky(kernelmatrix(kernel(theta), Xu))` WITHOUT jitter
(helper_functions/derivative_helper.jl:24-25).  This probe (NumPy only, CPU) takes the first minibatch of the run exactly as
the driver does (Probit moment matching, q(v), q(w)) and then
  1. counts how far a plain Cholesky of the un-jittered K_uu gets (first non-positive pivot) and how many pivots fall under
     PositiveFactorizations' threshold 10 M eps max|diag|;
  2. differentiates the objective in forward mode through three readings of "a Cholesky that does not fail" -- sub-threshold
     pivot dropped (unit diagonal), sub-threshold pivot raised to the threshold, |pivot| with no threshold -- and through
     the jittered factor the device uses, and prints the four gradients side by side.
Usage: python tools/banana_gradient_probe.py"""
import os
import sys

import numpy as np
from scipy.linalg import solve_triangular

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocessnode_amd.train import probit_marginal  # noqa: E402  (closed-form Probit marginal, host arithmetic)


def softplus(x):
    return np.logaddexp(0.0, x)


def gram(A, B, s2, ell):
    d = (A[:, None, :] - B[None, :, :]) / ell
    return s2 * np.exp(-0.5 * np.sum(d * d, axis=2))


def gram_with_derivs(A, B, s2, ell):
    """K and dK/d(s2, ell_1, ell_2)."""
    diff = A[:, None, :] - B[None, :, :]
    K = gram(A, B, s2, ell)
    dK = [K / s2] + [K * diff[:, :, d] ** 2 / ell[d] ** 3 for d in range(A.shape[1])]
    return K, np.stack(dK)


def chol_forward(A, dA, mode, tol):
    """Right-looking Cholesky of A with forward-mode derivatives dA[p]; `mode` says what a pivot <= tol becomes."""
    A, dA = A.copy(), dA.copy()
    M = A.shape[0]
    L, dL = np.zeros_like(A), np.zeros_like(dA)
    low = 0
    for j in range(M):
        p, dp = A[j, j], dA[:, j, j]
        if mode == "jitter" and p <= 0:
            raise np.linalg.LinAlgError(f"pivot {j}")
        if abs(p) <= tol and mode in ("drop", "raise"):
            low += 1
            if mode == "drop":                       # column dropped, unit diagonal: the point leaves the inducing set
                L[j, j] = 1.0
                continue
            p, dp = tol, np.zeros_like(dp)           # pivot raised to the threshold
        sgn = 1.0 if p > 0 else -1.0
        s = np.sqrt(abs(p))
        ds = 0.5 * sgn * dp / s
        L[j, j], dL[:, j, j] = s, ds
        col, dcol = A[j + 1:, j], dA[:, j + 1:, j]
        L[j + 1:, j] = col * (sgn / s)
        dL[:, j + 1:, j] = dcol * (sgn / s) - col[None, :] * (sgn / s ** 2) * ds[:, None]
        l, dl = L[j + 1:, j], dL[:, j + 1:, j]
        A[j + 1:, j + 1:] -= sgn * np.outer(l, l)
        for q in range(dA.shape[0]):
            o = np.outer(dl[q], l)
            dA[q, j + 1:, j + 1:] -= sgn * (o + o.T)
    return L, dL, low


def main():
    fix = np.load(os.path.join(ROOT, "tests", "golden", "banana_fixture.npz"))
    data, Xu = fix["data"], fix["Xu"]
    X, lab = data[:, :2], np.where(data[:, 2] < 0, 0.0, data[:, 2])
    xb, yb = X[:200], lab[:200]
    M = Xu.shape[0]
    theta = np.log(np.expm1(np.ones(3)))
    s2, ell = softplus(theta[0]), softplus(theta[1:])
    # ---- first minibatch, as perform_inference_classification does it ----
    a0 = b0 = 0.01
    w0 = a0 / b0
    Kuf = gram(Xu, xb, s2, ell)
    mf, vf = probit_marginal(yb, Kuf.T @ np.zeros(M), 1.0 / w0)
    Lam = np.eye(M) / 50.0 + w0 * Kuf @ Kuf.T
    Sigma = np.linalg.inv(Lam)
    mu = Sigma @ (w0 * Kuf @ mf)
    R = Sigma + np.outer(mu, mu)
    Kuu = gram(Xu, Xu, s2, ell)
    Lj = np.linalg.cholesky(Kuu + 1e-8 * np.eye(M))
    al = solve_triangular(Lj, Kuf, lower=True)
    I1 = s2 - np.sum(al * al, axis=0)
    I2 = mf ** 2 + vf - 2 * mf * (Kuf.T @ mu) + np.einsum("in,ij,jn->n", Kuf, R, Kuf)
    w = (a0 + 100.0) / (b0 + 0.5 * np.sum(I1 + I2))
    # ---- 1. how far does a plain Cholesky get without jitter ----
    ev = np.linalg.eigvalsh(Kuu)
    tol = 10 * M * np.finfo(float).eps * np.max(np.abs(np.diag(Kuu)))
    A = Kuu.copy()
    first_fail = None
    for j in range(M):
        if A[j, j] <= 0:
            first_fail = j + 1
            break
        l = A[j + 1:, j] / np.sqrt(A[j, j])
        A[j + 1:, j + 1:] -= np.outer(l, l)
    print(f"K_uu at theta_init, M = {M}: {np.sum(ev < 0)} negative eigenvalues (min {ev.min():.3e}), "
          f"{np.sum(ev < tol)} below 10 M eps max|diag| = {tol:.3e}; plain Cholesky stops at minor {first_fail}")
    # ---- 2. gradients of neg_log_backwardmess_fast at the first minibatch ----
    Kuf, dKuf = gram_with_derivs(Xu, xb, s2, ell)
    Kuu, dKuu = gram_with_derivs(Xu, Xu, s2, ell)
    # terms that do not involve K_uu: -w/2 k_nn - w/2 k'Rk + w y k'mu   (d k_nn / d s2 = 1)
    base = np.array([np.sum(-0.5 * w * np.einsum("in,ij,jn->n", 2 * dKuf[p], R, Kuf) + w * mf * (dKuf[p].T @ mu)) for p in range(3)])
    base[0] += -0.5 * w * len(yb)
    sig = 1.0 / (1.0 + np.exp(-theta))                                  # chain rule through softplus
    rows = []
    for mode, jit in (("jitter", 1e-8), ("drop", 0.0), ("raise", 0.0), ("abs", 0.0)):
        L, dL, low = chol_forward(Kuu + jit * np.eye(M), dKuu, mode, tol)
        al = solve_triangular(L, Kuf, lower=True)
        t1 = np.array([np.sum(al * solve_triangular(L, dKuf[p] - dL[p] @ al, lower=True)) for p in range(3)]) * w
        g = -(base + t1) * sig
        rows.append((mode, low, 0.5 * w * np.sum(al * al), g))
    print(f"objective term w/2 sum |L^-1 k_n|^2 and d(neg_log_backwardmess_fast)/d(theta_raw) at minibatch 1 (w = {w:.4f}):")
    for mode, low, t, g in rows:
        print(f"  {mode:7s} sub-threshold pivots {low:3d}   term {t: .6e}   grad {np.array2string(g, precision=6)}")
    ref = rows[0][3]
    for mode, low, t, g in rows[1:]:
        print(f"  {mode:7s} vs jitter: relative gradient difference {np.linalg.norm(g - ref) / np.linalg.norm(ref):.3e}")


if __name__ == "__main__":
    main()
