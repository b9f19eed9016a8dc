"""CPU oracle (NumPy, FP64) for the sparse-GP node's VMP hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of biaslab/GaussianProcessNode's UniSGP / MultiSGP
message rules.  It exists to CHECK the HIP path: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product (`gaussianprocessnode_amd`) never
does, and fails loudly when the HIP library is missing.

Parity status
-------------
The reference is Julia and cannot be executed in this pipeline (no `julia` binary).
The oracle is therefore pinned by the reference's own saved artefacts (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference/savefiles and /root/reference/data):
  * kin40k: SMSE(ytest, K(X*,Xu; theta_opt) mu_v) == savefiles/SMSE_kin40k.jld (0.08343114...)
  * banana: 125 test errors / rate 0.0961538 from savefiles/{qv,Xu,params_optimal}_banana.jld
  * kin40k closed-form q(v) vs the saved posterior (loose, theta drifted in the last epoch)
and by every analytic identity of the reference's GPtest.jl (restated in tests/test_oracle_*.py).
Third-party pieces the reference calls but does not contain (KernelFunctions.jl SE kernel,
ReactiveMP.jl Gaussian/Gamma/Wishart products, `ghcubature`, `srcubature`) are restated from
their published definitions; the cubature rules are "parity unpinned" (no fixture pins them
beyond the loose Monte-Carlo checks of GPtest.jl:141-143,380-382).

Conventions
-----------
NumPy arrays are C-ordered with points along axis 0: X is (N, D), Xu is (M, D).  In memory that
is the Julia layout of the reference (column-major D x N), so the same buffers go through the C ABI.
`Kuf` is returned as (M, N) (column n = the reference's `Psi1_trans` for point n).

All `file:line` citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.special import digamma
from scipy.linalg import cholesky, solve_triangular

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------
# kernel  (KernelFunctions.jl: theta[1] * with_lengthscale(SEKernel(), ell); call sites
#          GPnode/UniSGPnode.jl:102,153,169,205-206; experiments/regression_kin40k.ipynb:108)
# --------------------------------------------------------------------------------------------
def softplus(x):
    """StatsFuns.softplus, used by the notebooks' kernel_gp (experiments/regression_kin40k.ipynb:108)."""
    x = np.asarray(x, dtype=np.float64)
    return np.logaddexp(0.0, x)


def invsoftplus(y):
    y = np.asarray(y, dtype=np.float64)
    return y + np.log(-np.expm1(-y))


def kernel_from_theta(theta, softplus_params: bool):
    """theta -> (sigma2, lengthscales).  `softplus_params=True` is the notebooks' parameterisation
    (experiments/regression_kin40k.ipynb:108); False is GPtest.jl:21 (theta used directly)."""
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    if softplus_params:
        theta = softplus(theta)
    return float(theta[0]), theta[1:].copy()


def _as2d(A):
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = A[:, None]
    return A


def kernelmatrix(sigma2: float, ell, A, B=None):
    """K[i,j] = sigma2 * exp(-0.5 * sum_d ((A[i,d]-B[j,d])/ell_d)^2)   (SE / ARD-SE kernel).

    Direct difference form (no |a|^2+|b|^2-2ab expansion) so that tiny distances keep full
    relative accuracy -- the HIP Gram kernels use the same form."""
    A = _as2d(A)
    B = A if B is None else _as2d(B)
    ell = np.broadcast_to(np.asarray(ell, dtype=np.float64).ravel(), (A.shape[1],)) \
        if np.size(ell) in (1, A.shape[1]) else None
    if ell is None:
        raise ValueError("lengthscale must be a scalar or have one entry per input dimension")
    As = A / ell
    Bs = B / ell
    d2 = np.zeros((As.shape[0], Bs.shape[0]))
    for d in range(As.shape[1]):
        diff = As[:, d][:, None] - Bs[:, d][None, :]
        d2 += diff * diff
    return sigma2 * np.exp(-0.5 * d2)


def kernelmatrix_diag(sigma2: float, A):
    return np.full(_as2d(A).shape[0], float(sigma2))


# --------------------------------------------------------------------------------------------
# per-point rules, literal restatement (GPnode/UniSGPnode.jl) -- used on small cases to show that
# the batched forms below are the same mathematics
# --------------------------------------------------------------------------------------------
def rule_v_point(x, mu_y, w_bar, Xu, sigma2, ell):
    """@rule UniSGP(:v) regression/classification, PointMass input (GPnode/UniSGPnode.jl:144-158,161-173):
    returns (xi_n, Lambda_n) of MvNormalWeightedMeanPrecision."""
    k = kernelmatrix(sigma2, ell, Xu, np.atleast_2d(x))[:, 0]     # :153 / :169
    Lam = w_bar * np.outer(k, k)                                    # :155 / :170
    xi = k * (mu_y * w_bar)                                         # :156 / :171
    return xi, Lam


def prod_fold(mu0, Sigma0, messages):
    """prod(GenericProd, left::Normal, right::BufferUniSGP) folded over all messages
    (GPnode/UniSGPnode.jl:62-73).  The prior arrives as mean/covariance
    (experiments/regression_kin40k.ipynb:148) and is converted to weighted-mean/precision by
    ReactiveMP before the first product.  Returns mu_v, Sigma_v, Uv (= chol(Sigma_v + mu mu^T).U)."""
    Lam = cholinv(np.asarray(Sigma0, dtype=np.float64))
    xi = Lam @ np.asarray(mu0, dtype=np.float64)
    for xi_n, Lam_n in messages:                                    # :63, N times
        Lam = Lam + Lam_n
        xi = xi + xi_n
    Sigma_v = cholinv(Lam)                                          # :66 mean_cov
    mu_v = Sigma_v @ xi
    Rv = Sigma_v + np.outer(mu_v, mu_v)                             # :67
    Uv = cholesky(Rv, lower=False)                                  # :68
    return mu_v, Sigma_v, Uv


def rule_w_point(x, mu_y, v_y, mu_v, Uv, KuuL, Xu, sigma2, ell):
    """@rule UniSGP(:w) PointMass input (GPnode/UniSGPnode.jl:196-216 regression, :219-238
    classification).  Returns (I1, I2); the message is GammaShapeRate(1.5, 0.5*(I1+I2))."""
    k = kernelmatrix(sigma2, ell, Xu, np.atleast_2d(x))[:, 0]       # :206
    k0 = sigma2                                                      # :205
    alpha = solve_triangular(KuuL, k, lower=True)                   # :208
    I1 = k0 - alpha @ alpha                                         # :209
    I2 = mu_y * mu_y + v_y - 2.0 * mu_y * (k @ mu_v)                # :211 / :234
    beta = Uv @ k                                                   # :212
    I2 += beta @ beta                                               # :213
    return I1, I2


def average_energy_point(I1, I2, w_bar, E_logw):
    """@average_energy UniSGP, PointMass input (GPnode/UniSGPnode.jl:337-359,363-387,411-436)."""
    return 0.5 * (I1 * w_bar - E_logw + LOG2PI + I2 * w_bar)


def rule_out_point(x, mu_v, w_bar, Xu, sigma2, ell):
    """@rule UniSGP(:out) PointMass input (GPnode/UniSGPnode.jl:96-104): NormalMeanPrecision(k^T mu_v, w_bar)."""
    k = kernelmatrix(sigma2, ell, Xu, np.atleast_2d(x))[:, 0]
    return float(k @ mu_v), float(w_bar)


# --------------------------------------------------------------------------------------------
# dense helpers standing in for FastCholesky.jl (cholinv / fastcholesky)
# --------------------------------------------------------------------------------------------
def cholinv(A):
    L = cholesky(np.asarray(A, dtype=np.float64), lower=True)
    Linv = solve_triangular(L, np.eye(L.shape[0]), lower=True)
    return Linv.T @ Linv


def chol_lower(A):
    return cholesky(np.asarray(A, dtype=np.float64), lower=True)


# --------------------------------------------------------------------------------------------
# batched restatement -- what the HIP kernels implement
# --------------------------------------------------------------------------------------------
@dataclass
class SuffStats:
    """Additive sufficient statistics of a set of points (eq. S)."""
    Psi2: np.ndarray        # (M, M)  K_uf diag(omega) K_uf^T
    b: np.ndarray           # (M, Do) K_uf (omega * mu_y)
    s_yy: np.ndarray        # (Do, Do) sum omega (mu_y mu_y^T + v_y)   (scalar as 1x1 for UniSGP)
    s_kk: float             # sum omega k(x,x)
    n: float                # number of nodes (sum over points of 1; cubature points of one node share 1)

    def __add__(self, o):
        return SuffStats(self.Psi2 + o.Psi2, self.b + o.b, self.s_yy + o.s_yy, self.s_kk + o.s_kk, self.n + o.n)


def kuu_and_chol(Xu, sigma2, ell, jitter=0.0):
    """Caller side of the hot path (experiments/regression_kin40k.ipynb:183-184; banana adds 1e-8 I,
    experiments/classification_banana.ipynb:163): K_uu and its lower Cholesky factor."""
    Kuu = kernelmatrix(sigma2, ell, Xu)
    if jitter:
        Kuu = Kuu + jitter * np.eye(Kuu.shape[0])
    return Kuu, chol_lower(Kuu)


def suff_stats(Xu, X, y_mean, y_var, sigma2, ell, omega=None, n_nodes=None) -> SuffStats:
    """(S): sums over points of the :v-rule messages (GPnode/UniSGPnode.jl:153-156,169-171)."""
    X = _as2d(X)
    Kuf = kernelmatrix(sigma2, ell, Xu, X)                          # (M, N)
    N = X.shape[0]
    om = np.ones(N) if omega is None else np.asarray(omega, dtype=np.float64)
    Y = np.asarray(y_mean, dtype=np.float64).reshape(N, -1)
    Psi2 = (Kuf * om) @ Kuf.T
    b = Kuf @ (Y * om[:, None])
    s_yy = (Y * om[:, None]).T @ Y
    if y_var is not None:
        yv = np.asarray(y_var, dtype=np.float64)
        if yv.ndim == 1:
            s_yy = s_yy + np.sum(om * yv) * np.eye(1)
        else:                                                       # (N, Do, Do) covariances
            s_yy = s_yy + np.tensordot(om, yv, axes=(0, 0))
    return SuffStats(Psi2, b, s_yy, float(sigma2 * om.sum()), float(N if n_nodes is None else n_nodes))


def v_update(stats: SuffStats, w_bar: float, mu0=None, Sigma0=None, Lambda0=None, xi0=None):
    """(V): q(v) from the prior and the summed messages (GPnode/UniSGPnode.jl:62-71).
    Returns mu_v, Sigma_v, Uv (upper, Uv^T Uv = Sigma_v + mu_v mu_v^T)."""
    if Lambda0 is None:
        Lambda0 = cholinv(Sigma0)
        xi0 = Lambda0 @ np.asarray(mu0, dtype=np.float64)
    Lam = Lambda0 + w_bar * stats.Psi2
    xi = xi0 + w_bar * stats.b[:, 0]
    Sigma_v = cholinv(Lam)
    mu_v = Sigma_v @ xi
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    Uv = cholesky(Rv, lower=False)
    return mu_v, Sigma_v, Uv


def w_stats_perpoint(Xu, X, y_mean, y_var, sigma2, ell, KuuL, mu_v, Uv):
    """(W): per-point I1_n (= Q_ff diagonal term) and I2_n  (GPnode/UniSGPnode.jl:196-238)."""
    X = _as2d(X)
    Kuf = kernelmatrix(sigma2, ell, Xu, X)
    alpha = solve_triangular(KuuL, Kuf, lower=True)
    I1 = sigma2 - np.sum(alpha * alpha, axis=0)
    y = np.asarray(y_mean, dtype=np.float64).ravel()
    vy = np.zeros_like(y) if y_var is None else np.asarray(y_var, dtype=np.float64).ravel()
    beta = Uv @ Kuf
    I2 = y * y + vy - 2.0 * y * (Kuf.T @ mu_v) + np.sum(beta * beta, axis=0)
    return I1, I2


def w_stats_trace(stats: SuffStats, KuuL, mu_v, Sigma_v):
    """(W'): sum_n I1_n and sum_n I2_n from the reduced statistics
    (trace identity used by the reference's own uncertain-input rule, GPnode/UniSGPnode.jl:189-190)."""
    A = solve_triangular(KuuL, stats.Psi2, lower=True)
    A = solve_triangular(KuuL, A.T, lower=True)                     # L^-1 Psi2 L^-T
    sum_I1 = stats.s_kk - np.trace(A)
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    sum_I2 = float(stats.s_yy[0, 0]) - 2.0 * float(stats.b[:, 0] @ mu_v) + float(np.sum(Rv * stats.Psi2))
    return float(sum_I1), float(sum_I2)


def gamma_update(a0, b0, n, sum_I1, sum_I2):
    """Product of the prior Gamma with n messages GammaShapeRate(1.5, r_i): shape a0 + n/2, rate b0 + sum r_i
    (GPnode/UniSGPnode.jl:215,237; confirmed by savefiles/qw_banana.jld)."""
    return a0 + 0.5 * n, b0 + 0.5 * (sum_I1 + sum_I2)


def gamma_mean_logmean(a, b):
    return a / b, float(digamma(a) - math.log(b))


def avg_energy_sum(n, sum_I1, sum_I2, w_bar, E_logw):
    """(E): sum over the n nodes of the average energy (GPnode/UniSGPnode.jl:337-359,411-436)."""
    return 0.5 * (w_bar * (sum_I1 + sum_I2) - n * E_logw + n * LOG2PI)


def predict_mean(Xu, Xstar, mu_v, sigma2, ell):
    """(P): batched :out rule (GPnode/UniSGPnode.jl:96-104; loop experiments/regression_kin40k.ipynb:288-304)."""
    return kernelmatrix(sigma2, ell, _as2d(Xstar), Xu) @ mu_v


def theta_objective(Xu, X, y, sigma2, ell, mu_v, Uv, w_bar, jitter=0.0):
    """neg_log_backwardmess_fast (helper_functions/derivative_helper.jl:23-39): the hyper-parameter objective."""
    _, L = kuu_and_chol(Xu, sigma2, ell, jitter)
    Kuf = kernelmatrix(sigma2, ell, Xu, _as2d(X))
    alpha = solve_triangular(L, Kuf, lower=True)
    beta = Uv @ Kuf
    y = np.asarray(y, dtype=np.float64).ravel()
    llh = np.sum(-0.5 * w_bar * sigma2 + 0.5 * w_bar * np.sum(alpha * alpha, axis=0)
                 - 0.5 * w_bar * np.sum(beta * beta, axis=0) + w_bar * y * (Kuf.T @ mu_v))
    return -float(llh)


@dataclass
class SweepResult:
    mu_v: np.ndarray
    Sigma_v: np.ndarray
    Uv: np.ndarray
    sum_I1: float
    sum_I2: float
    energy: float
    KuuL: np.ndarray
    stats: SuffStats
    qw: Optional[Tuple[float, float]] = None


def vmp_sweep(Xu, X, y_mean, y_var, sigma2, ell, w_bar, E_logw=None, jitter=0.0,
              mu0=None, Sigma0=None, Lambda0=None, xi0=None, stats: Optional[SuffStats] = None) -> SweepResult:
    """One VMP sweep as BASELINE.md defines it: K_uu + L, K_uf, statistics, q(v) incl. Sigma_v and Uv,
    sum I1 / sum I2, summed average energy.  `stats` may be passed in already reduced (multi-GPU path)."""
    Xu = _as2d(Xu)
    _, L = kuu_and_chol(Xu, sigma2, ell, jitter)
    if stats is None:
        stats = suff_stats(Xu, X, y_mean, y_var, sigma2, ell)
    mu_v, Sigma_v, Uv = v_update(stats, w_bar, mu0, Sigma0, Lambda0, xi0)
    s1, s2 = w_stats_trace(stats, L, mu_v, Sigma_v)
    if E_logw is None:
        E_logw = math.log(w_bar)
    U = avg_energy_sum(stats.n, s1, s2, w_bar, E_logw)
    return SweepResult(mu_v, Sigma_v, Uv, s1, s2, U, L, stats)


# --------------------------------------------------------------------------------------------
# metrics and batching (helper_functions/gp_helperfunction.jl:137-158)
# --------------------------------------------------------------------------------------------
def split2batch(x, y, batch_size):
    """:137-142"""
    xb = [x[i:i + batch_size] for i in range(0, len(x), batch_size)]
    yb = [y[i:i + batch_size] for i in range(0, len(y), batch_size)]
    return xb, yb


def SMSE(y_true, y_approx):
    """:145-149 -- Julia `var` is the unbiased (ddof=1) sample variance."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_approx = np.asarray(y_approx, dtype=np.float64)
    mse = np.linalg.norm(y_true - y_approx) ** 2 / y_true.size
    return mse / np.var(y_true, ddof=1)


def num_error(ytrue, y):
    """:152-154"""
    return float(np.sum(np.abs(np.asarray(y, dtype=np.float64) - np.asarray(ytrue, dtype=np.float64))))


def error_rate(ytrue, y):
    """:156-158"""
    return num_error(ytrue, y) / len(ytrue)


# --------------------------------------------------------------------------------------------
# cubature rules (ReactiveMP.jl, not vendored by the reference -> restated from their published
# definitions; parity unpinned, see header)
# --------------------------------------------------------------------------------------------
def ghcubature_1d(p: int, m: float, P: float):
    """ReactiveMP `ghcubature(p)` for a univariate N(m, P): points m + sqrt(2P) x_i, weights w_i/sqrt(pi).
    Used by GPnode/UniSGPnode.jl:11-33 through getweights/getpoints."""
    xs, ws = np.polynomial.hermite.hermgauss(p)
    return m + math.sqrt(2.0 * P) * xs, ws / math.sqrt(math.pi)


def srcubature(m, P):
    """ReactiveMP `srcubature()` (spherical-radial cubature with a central point): 2d+1 points
    m + sqrt(d+1) L (+/- e_i) with weight 1/(2(d+1)) each, and m with weight 1/(d+1); L = chol(P).L.
    Used by GPnode/MultiSGPnode.jl:11-35."""
    m = np.asarray(m, dtype=np.float64)
    d = m.size
    L = chol_lower(np.asarray(P, dtype=np.float64))
    pts = []
    for i in range(d):
        pts.append(m + math.sqrt(d + 1.0) * L[:, i])
    for i in range(d):
        pts.append(m - math.sqrt(d + 1.0) * L[:, i])
    pts.append(m.copy())
    w = np.full(2 * d + 1, 1.0 / (2.0 * (d + 1.0)))
    w[-1] = 1.0 / (d + 1.0)
    return np.stack(pts), w


def psi_statistics(Xu, points, weights, sigma2, ell):
    """approximate_kernel_expectation of k(x,x), K(Xu,x), K(Xu,x)K(x,Xu)
    (GPnode/UniSGPnode.jl:11-33, GPnode/MultiSGPnode.jl:11-35): returns Psi0, Psi1 (M,), Psi2 (M,M)."""
    pts = _as2d(points)
    K = kernelmatrix(sigma2, ell, Xu, pts)                          # (M, S)
    w = np.asarray(weights, dtype=np.float64)
    return float(sigma2 * w.sum()), K @ w, (K * w) @ K.T


# --------------------------------------------------------------------------------------------
# MultiSGP (GPnode/MultiSGPnode.jl), batched over the steps of a sequence
# --------------------------------------------------------------------------------------------
def multi_rule_v(Psi1, Psi2, mu_y, W):
    """@rule MultiSGP(:v) (GPnode/MultiSGPnode.jl:290-308): xi = vcat_d(Psi1 * (mu_y^T W)_d), Lambda = kron(W, Psi2)."""
    row = np.asarray(mu_y) @ W                                      # :307  mul_A_B!(mu_y', W)
    xi = np.concatenate([Psi1 * row[d] for d in range(len(row))])
    return xi, np.kron(W, Psi2)                                     # :306


def multi_rule_in_logpdf(Xu, sigma2, ell, mu_y, mu_v, Sigma_v, W, Kuu_inv):
    """@rule MultiSGP(:in) with a Gaussian or point-mass output (GPnode/MultiSGPnode.jl:162-184, :186-208): the log-pdf closure
    x -> -1/2 tr(W) (k(x,x) - sum(Kuu^-1 .* k k')) + sumdiagV . k - 1/2 sum(k k' .* sumRvblk_W),  k = K(Xu, x),
    sumdiagV = sum_d V[dM:(d+1)M, d] with V = mu_v mu_y' W (sum_diagonal_M, helper_functions/derivative_helper.jl:117-120),
    sumRvblk_W = sum_ij Rv_blk[i][j] W[i][j] (create_blockmatrix, helper_functions/gp_helperfunction.jl:133-135)."""
    Xu = np.atleast_2d(np.asarray(Xu, dtype=np.float64))
    M = Xu.shape[0]
    mu_y = np.asarray(mu_y, dtype=np.float64).ravel()
    D = len(mu_y)
    W = np.atleast_2d(np.asarray(W, dtype=np.float64))
    mu_v = np.asarray(mu_v, dtype=np.float64).ravel()
    Rv = np.asarray(Sigma_v, dtype=np.float64) + np.outer(mu_v, mu_v)                   # :176
    V = np.outer(mu_v, mu_y) @ W                                                        # :177
    sumdiagV = sum(V[d * M:(d + 1) * M, d] for d in range(D))                           # :178
    sumRvblk_W = sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * W[i, j] for i in range(D) for j in range(D))   # :179
    trW = float(np.trace(W))

    def log_backwardmess(x):
        x = np.asarray(x, dtype=np.float64).reshape(1, -1)
        k = kernelmatrix(sigma2, ell, Xu, x)[:, 0]
        Psi0 = float(kernelmatrix(sigma2, ell, x)[0, 0])
        Psi2 = np.outer(k, k)
        return float(-0.5 * trW * (Psi0 - np.sum(Kuu_inv * Psi2)) + sumdiagV @ k - 0.5 * np.sum(Psi2 * sumRvblk_W))   # :181
    return log_backwardmess


def multi_rule_theta_logpdf(Xu, kernel, pts, wts, mu_y, mu_v, Sigma_v, W):
    """@rule MultiSGP(:theta) (GPnode/MultiSGPnode.jl:447-466): theta -> -1/2 tr(W I1(theta)) + mu_y' W kron(C, Psi1(theta)) mu_v
    - 1/2 tr(kron(W, Psi2(theta)) Rv), I1 = kron(C, Psi0 - tr(Kuu^-1(theta) Psi2(theta))), Psi2 carries + 1e-7 I (:458), Kuu no
    jitter (:455).  `kernel`: theta -> (sigma2, lengthscales); (pts, wts): the cubature rule of q_in."""
    Xu = np.atleast_2d(np.asarray(Xu, dtype=np.float64))
    M = Xu.shape[0]
    mu_y = np.asarray(mu_y, dtype=np.float64).ravel()
    D = len(mu_y)
    W = np.atleast_2d(np.asarray(W, dtype=np.float64))
    mu_v = np.asarray(mu_v, dtype=np.float64).ravel()
    Rv = np.asarray(Sigma_v, dtype=np.float64) + np.outer(mu_v, mu_v)                   # :450
    C = np.eye(D)

    def log_backwardmess(theta):
        sigma2, ell = kernel(theta)
        Psi0, Psi1, Psi2 = psi_statistics(Xu, pts, wts, sigma2, ell)
        Psi2 = Psi2 + 1e-7 * np.eye(M)                                                  # :458
        Kinv = cholinv(kernelmatrix(sigma2, ell, Xu))                                   # :455
        I1 = np.kron(C, np.atleast_2d(Psi0 - np.trace(Kinv @ Psi2)))                    # :460
        Psi1_tilde = np.kron(C, np.asarray(Psi1)[None, :])                              # :461
        Psi3 = np.kron(W, Psi2)                                                         # :462
        return float(-0.5 * np.trace(W @ I1) + mu_y @ W @ Psi1_tilde @ mu_v - 0.5 * np.trace(Psi3 @ Rv))   # :463
    return log_backwardmess


def multi_rule_w(Psi0, Psi1, Psi2, mu_y, Sigma_y, mu_v, Sigma_v, Kuu_inv):
    """@rule MultiSGP(:w) (GPnode/MultiSGPnode.jl:367-405,407-444): inverse scale I1 + I2 of WishartFast(D+2, .)."""
    D = len(mu_y)
    M = Psi1.shape[0]
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    I1 = (Psi0 - np.trace(Kuu_inv @ Psi2)) * np.eye(D)              # :391-392
    E = np.array([Psi1 @ mu_v[d * M:(d + 1) * M] for d in range(D)])  # :395
    Psi4 = np.array([[np.sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * Psi2.T) for j in range(D)]
                     for i in range(D)])                            # :397
    tmp = np.outer(mu_y, E)
    Ry = np.outer(mu_y, mu_y) + (0.0 if Sigma_y is None else Sigma_y)
    return Psi4 + Ry - (tmp + tmp.T) + I1                           # :398-404


def multi_rule_out(Psi1, mu_v, D):
    """@rule MultiSGP(:out) (GPnode/MultiSGPnode.jl:90-104): mean_d = Psi1 . mu_v^(d)."""
    M = Psi1.shape[0]
    return np.array([Psi1 @ mu_v[d * M:(d + 1) * M] for d in range(D)])


def multi_average_energy(Psi0, Psi1, Psi2, mu_y, Sigma_y, mu_v, Sigma_v, W, E_logdetW, Kuu_inv):
    """@average_energy MultiSGP (GPnode/MultiSGPnode.jl:544-632)."""
    D = len(mu_y)
    M = Psi1.shape[0]
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    V = np.outer(mu_v, mu_y) @ W                                    # (DM, D)
    sumdiagV = sum(V[d * M:(d + 1) * M, d] for d in range(D))       # derivative_helper.jl:119-122
    sumRvblk_W = sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * W[i, j] for i in range(D) for j in range(D))
    Ry = np.outer(mu_y, mu_y) + (0.0 if Sigma_y is None else Sigma_y)
    return (0.5 * D * LOG2PI - 0.5 * E_logdetW + 0.5 * np.trace(W @ Ry)
            + 0.5 * np.trace(W) * (Psi0 - np.sum(Kuu_inv * Psi2)) - np.sum(sumdiagV * Psi1)
            + 0.5 * np.sum(Psi2 * sumRvblk_W))


def multi_log_backward_in(x, Xu, sigma2, ell, mu_y, mu_v, Sigma_v, W, Kuu_inv):
    """log-pdf closure of @rule MultiSGP(:in) (GPnode/MultiSGPnode.jl:162-184) evaluated at x."""
    D = len(mu_y)
    M = _as2d(Xu).shape[0]
    k = kernelmatrix(sigma2, ell, Xu, np.atleast_2d(x))[:, 0]
    Psi2 = np.outer(k, k)
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    V = np.outer(mu_v, mu_y) @ W
    sumdiagV = sum(V[d * M:(d + 1) * M, d] for d in range(D))
    sumRvblk_W = sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * W[i, j] for i in range(D) for j in range(D))
    return (-0.5 * np.trace(W) * (sigma2 - np.sum(Kuu_inv * Psi2)) + np.sum(sumdiagV * k)
            - 0.5 * np.sum(Psi2 * sumRvblk_W))


@dataclass
class MultiStats:
    """Additive statistics of a MultiSGP sequence (eq. M)."""
    Psi2: np.ndarray        # (M, M)   sum_t Psi2_t
    B: np.ndarray           # (M, Do)  sum_t Psi1_t mu_y,t^T
    Ryy: np.ndarray         # (Do, Do) sum_t (mu_y mu_y^T + Sigma_y)
    s_kk: float             # sum_t Psi0_t
    n: float


def multi_suff_stats(Xu, points, weights, Y, Sigma_y, sigma2, ell) -> MultiStats:
    """points: (T, S, Din) cubature points per step, weights: (T, S); Y: (T, Do)."""
    points = np.asarray(points, dtype=np.float64)
    T, S, _ = points.shape
    K = kernelmatrix(sigma2, ell, Xu, points.reshape(T * S, -1))    # (M, T*S)
    w = np.asarray(weights, dtype=np.float64).reshape(T * S)
    Psi2 = (K * w) @ K.T
    Psi1 = (K * w).reshape(K.shape[0], T, S).sum(axis=2)            # (M, T)
    B = Psi1 @ np.asarray(Y, dtype=np.float64)
    Ryy = np.asarray(Y).T @ np.asarray(Y)
    if Sigma_y is not None:
        Ryy = Ryy + np.sum(np.asarray(Sigma_y), axis=0)
    return MultiStats(Psi2, B, Ryy, float(sigma2 * w.sum()), float(T))


def multi_v_update(ms: MultiStats, W, Lambda0, xi0):
    """Summed :v messages + prior: Lambda = Lambda0 + kron(W, Psi2), xi = xi0 + vec(B W) output-major
    (GPnode/MultiSGPnode.jl:306-307)."""
    Lam = Lambda0 + np.kron(W, ms.Psi2)
    xi = xi0 + (ms.B @ W).T.reshape(-1)
    Sigma_v = cholinv(Lam)
    return Sigma_v @ xi, Sigma_v


def multi_w_update(ms: MultiStats, mu_v, Sigma_v, Kuu_inv):
    """Sum over steps of the :w messages' inverse scales (GPnode/MultiSGPnode.jl:391-404)."""
    D = ms.B.shape[1]
    M = ms.Psi2.shape[0]
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    I1 = (ms.s_kk - np.sum(Kuu_inv * ms.Psi2)) * np.eye(D)
    EY = np.array([[mu_v[d * M:(d + 1) * M] @ ms.B[:, e] for d in range(D)] for e in range(D)])  # sum_t mu_y E^T
    Psi4 = np.array([[np.sum(Rv[i * M:(i + 1) * M, j * M:(j + 1) * M] * ms.Psi2) for j in range(D)]
                     for i in range(D)])
    return I1 + ms.Ryy - EY - EY.T + Psi4
