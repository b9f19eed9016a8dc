"""The reference's GPtest.jl identities restated against the oracle, plus the
batched-vs-per-point equivalences the HIP design rests on (sum of N rank-1 messages == one SYRK).

Sizes follow GPtest.jl:14-35 (Nu = 10 1-D inducing points, Nu_2d = 25 grid).  The reference draws
unseeded random inputs; a seeded generator is used here, the identities hold for any draw.
"""
import math

import numpy as np
import pytest
from scipy.special import digamma

from oracle import sgp_oracle as O

THETA = np.array([1.0, 1.0])                      # GPtest.jl:16
S2, ELL = O.kernel_from_theta(THETA, softplus_params=False)
XU = np.arange(1.0, 11.0)[:, None]                # GPtest.jl:19
XU2 = np.array([[i, j] for j in range(1, 6) for i in range(1, 6)], dtype=np.float64)  # GPtest.jl:20


@pytest.fixture()
def uni():
    rng = np.random.default_rng(7)
    mu_v = np.sin(rng.random(10))                 # GPtest.jl:117
    Sigma_v = np.eye(10)
    Rv = Sigma_v + np.outer(mu_v, mu_v)
    Kuu, L = O.kuu_and_chol(XU, S2, ELL)
    Uv = np.linalg.cholesky(Rv).T
    return dict(mu_v=mu_v, Sigma_v=Sigma_v, Rv=Rv, Kuu=Kuu, L=L, Uv=Uv, Kinv=np.linalg.inv(Kuu))


def test_rule_out_pointmass(uni):
    """GPtest.jl:163-169"""
    m, w = O.rule_out_point(1.0, uni["mu_v"], 1.0, XU, S2, ELL)
    Psi1 = O.kernelmatrix(S2, ELL, np.array([[1.0]]), XU)
    assert math.isclose(m, float((Psi1 @ uni["mu_v"])[0]), rel_tol=1e-13)
    assert w == 1.0


def test_rule_v_pointmass(uni):
    """GPtest.jl:194-216: mean = inv(Psi2) Psi1' y, cov = inv(w Psi2) -- on the (singular) rank-1 message the
    reference compares through cholinv; here the natural parameters themselves are compared."""
    w = 1.0
    xi, Lam = O.rule_v_point(1.0, 2.0, w, XU, S2, ELL)
    k = O.kernelmatrix(S2, ELL, XU, np.array([[1.0]]))[:, 0]
    np.testing.assert_allclose(Lam, w * np.outer(k, k), rtol=1e-14)
    np.testing.assert_allclose(xi, w * 2.0 * k, rtol=1e-14)


@pytest.mark.parametrize("v_y", [0.0, 4.0])
def test_rule_w_pointmass(uni, v_y):
    """GPtest.jl:231-253: rate = 0.5 (I1 + I2), I1 = Psi0 - tr(Kuu^-1 Psi2), I2 = y^2 (+v) - 2 y Psi1 mu + tr(Rv Psi2)."""
    y = 2.0 if v_y == 0.0 else 1.0
    I1, I2 = O.rule_w_point(1.0, y, v_y, uni["mu_v"], uni["Uv"], uni["L"], XU, S2, ELL)
    k = O.kernelmatrix(S2, ELL, XU, np.array([[1.0]]))[:, 0]
    Psi2 = np.outer(k, k)
    I1_gt = S2 - np.trace(uni["Kinv"] @ Psi2)
    I2_gt = y * y + v_y - 2 * y * (k @ uni["mu_v"]) + np.trace(uni["Rv"] @ Psi2)
    assert math.isclose(I1, I1_gt, rel_tol=1e-9, abs_tol=1e-11)
    assert math.isclose(I2, I2_gt, rel_tol=1e-12)


def test_average_energy_pointmass(uni):
    """GPtest.jl:295-308 (Gamma w) and the PointMass-w variant (GPnode/UniSGPnode.jl:411-436)."""
    a, b = 1.0, 1.0
    w_bar, E_logw = O.gamma_mean_logmean(a, b)
    assert math.isclose(E_logw, digamma(1.0))
    I1, I2 = O.rule_w_point(1.0, 2.0, 0.0, uni["mu_v"], uni["Uv"], uni["L"], XU, S2, ELL)
    U = O.average_energy_point(I1, I2, w_bar, E_logw)
    U_gt = 0.5 * math.log(2 * math.pi) - 0.5 * E_logw + 0.5 * w_bar * (I1 + I2)
    assert math.isclose(U, U_gt, rel_tol=1e-14)
    U5 = O.average_energy_point(I1, I2, 5.0, math.log(5.0))
    assert math.isclose(U5, 0.5 * math.log(2 * math.pi) - 0.5 * math.log(5.0) + 2.5 * (I1 + I2), rel_tol=1e-14)


def test_theta_objective_matches_bruteforce(uni):
    """GPtest.jl:50-75 (derivative helper vs brute-force sum)."""
    xdata = np.arange(-5.0, 6.0)
    ydata = np.sin(xdata ** 2 - 1) + np.cos(xdata)
    w = 1.0
    gt = 0.0
    for x, y in zip(xdata, ydata):
        k = O.kernelmatrix(S2, ELL, XU, np.array([[x]]))[:, 0]
        gt += -0.5 * w * (S2 + np.trace(np.outer(k, k) @ (uni["Rv"] - uni["Kinv"]))) + w * y * (k @ uni["mu_v"])
    got = O.theta_objective(XU, xdata[:, None], ydata, S2, ELL, uni["mu_v"], uni["Uv"], w)
    assert math.isclose(-gt, got, rel_tol=1e-7, abs_tol=1e-6)


# ---------------------------------------------------------------- batched == per-point
@pytest.mark.parametrize("N,M,D,w_bar,with_var", [(37, 10, 1, 100.0, False), (64, 25, 2, 3.0, True),
                                                  (5, 12, 3, 1e4, False), (1, 4, 1, 1.0, True)])
def test_batched_equals_per_point_fold(N, M, D, w_bar, with_var):
    rng = np.random.default_rng(N * 1000 + M)
    X = rng.uniform(-2, 2, (N, D))
    Xu = rng.uniform(-2, 2, (M, D))
    y = rng.normal(size=N)
    vy = rng.uniform(0.1, 1.0, N) if with_var else None
    s2, ell = 0.7, rng.uniform(0.8, 2.0, D)
    mu0 = rng.normal(size=M) * 0.1
    A = rng.normal(size=(M, M))
    Sigma0 = A @ A.T / M + np.eye(M)

    msgs = [O.rule_v_point(X[n], y[n], w_bar, Xu, s2, ell) for n in range(N)]
    mu_pp, Sig_pp, Uv_pp = O.prod_fold(mu0, Sigma0, msgs)

    res = O.vmp_sweep(Xu, X, y, vy, s2, ell, w_bar, jitter=1e-8, mu0=mu0, Sigma0=Sigma0)
    np.testing.assert_allclose(res.mu_v, mu_pp, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.Sigma_v, Sig_pp, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.Uv, Uv_pp, rtol=1e-8, atol=1e-11)

    # (W) per point, (W') trace form and the literal rule agree
    I1, I2 = O.w_stats_perpoint(Xu, X, y, vy, s2, ell, res.KuuL, res.mu_v, res.Uv)
    lit = [O.rule_w_point(X[n], y[n], 0.0 if vy is None else vy[n], res.mu_v, res.Uv, res.KuuL, Xu, s2, ell)
           for n in range(N)]
    np.testing.assert_allclose(I1, [t[0] for t in lit], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(I2, [t[1] for t in lit], rtol=1e-10)
    # sum I1 = s_kk - tr(Kuu^-1 Psi2) cancels to ~cond(Kuu)*eps of s_kk: compare on that scale
    assert abs(res.sum_I1 - I1.sum()) <= 1e-7 * res.stats.s_kk
    assert math.isclose(res.sum_I2, I2.sum(), rel_tol=1e-9)

    # (E) summed average energy
    E_logw = math.log(w_bar)
    U_lit = sum(O.average_energy_point(a, b, w_bar, E_logw) for a, b in lit)
    assert math.isclose(res.energy, U_lit, rel_tol=1e-8)

    # statistics are additive over shards (the multi-GPU contract)
    cut = N // 2
    if 0 < cut < N:
        s_a = O.suff_stats(Xu, X[:cut], y[:cut], None if vy is None else vy[:cut], s2, ell)
        s_b = O.suff_stats(Xu, X[cut:], y[cut:], None if vy is None else vy[cut:], s2, ell)
        s_ab = s_a + s_b
        np.testing.assert_allclose(s_ab.Psi2, res.stats.Psi2, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(s_ab.b, res.stats.b, rtol=1e-12, atol=1e-14)
        assert math.isclose(s_ab.s_yy[0, 0], res.stats.s_yy[0, 0], rel_tol=1e-13)


def test_sequential_minibatch_carry_equals_full_batch():
    """Posterior->prior carry over minibatches (experiments/regression_kin40k.ipynb:203-212) is the same sum."""
    rng = np.random.default_rng(3)
    N, M, D = 120, 16, 2
    X, Xu, y = rng.uniform(-2, 2, (N, D)), rng.uniform(-2, 2, (M, D)), rng.normal(size=N)
    s2, ell, w = 1.3, np.array([1.0, 1.5]), 50.0
    mu, Sig = np.zeros(M), 50.0 * np.eye(M)
    xb, yb = O.split2batch(X, y, 50)
    assert [len(b) for b in xb] == [50, 50, 20]
    for xi, yi in zip(xb, yb):
        r = O.vmp_sweep(Xu, xi, yi, None, s2, ell, w, mu0=mu, Sigma0=Sig)
        mu, Sig = r.mu_v, r.Sigma_v
    full = O.vmp_sweep(Xu, X, y, None, s2, ell, w, mu0=np.zeros(M), Sigma0=50.0 * np.eye(M))
    np.testing.assert_allclose(mu, full.mu_v, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(Sig, full.Sigma_v, rtol=1e-8, atol=1e-11)


# ---------------------------------------------------------------- cubature (parity unpinned, loose)
def test_ghcubature_psi_statistics_vs_monte_carlo():
    """GPtest.jl:127-143 (atol 1e-4 / 0.05 / 0.05)."""
    rng = np.random.default_rng(0)
    pts, w = O.ghcubature_1d(21, 0.0, 1.0)
    assert math.isclose(w.sum(), 1.0, rel_tol=1e-12)
    P0, P1, P2 = O.psi_statistics(XU, pts[:, None], w, S2, ELL)
    xs = rng.normal(size=20000)
    K = O.kernelmatrix(S2, ELL, XU, xs[:, None])
    assert abs(P0 - 1.0) < 1e-4
    assert np.abs(P1 - K.mean(axis=1)).max() < 0.05
    assert np.abs(P2 - (K @ K.T) / xs.size).max() < 0.05


def test_srcubature_psi_statistics_vs_monte_carlo():
    """GPtest.jl:366-382 (Psi0 exact, atol 0.08 / 0.3)."""
    rng = np.random.default_rng(1)
    m, P = np.array([1.0, 2.7]), np.eye(2)
    pts, w = O.srcubature(m, P)
    assert pts.shape == (5, 2)
    P0, P1, P2 = O.psi_statistics(XU2, pts, w, S2, ELL)
    assert P0 == 1.0
    xs = rng.multivariate_normal(m, P, size=20000)
    K = O.kernelmatrix(S2, ELL, XU2, xs)
    assert np.abs(P1 - K.mean(axis=1)).max() < 0.08
    assert np.abs(P2 - (K @ K.T) / len(xs)).max() < 0.3
    # the rule reproduces mean and covariance exactly
    np.testing.assert_allclose(w @ pts, m, atol=1e-14)
    np.testing.assert_allclose((pts - m).T @ ((pts - m) * w[:, None]), P, atol=1e-13)


# ---------------------------------------------------------------- MultiSGP (GPtest.jl:352-539)
@pytest.fixture()
def multi():
    rng = np.random.default_rng(11)
    D, M = 2, 25
    mu_y, Sigma_y = np.array([0.5, 1.4]), np.eye(2)
    m_in, P_in = np.array([1.0, 2.7]), np.eye(2)
    mu_v = np.sin(rng.random(D * M))
    Sigma_v = np.eye(D * M)
    W = 10 * 50.0 * np.eye(2)                        # mean of Wishart(10, 50 I)  GPtest.jl:356
    Kuu = O.kernelmatrix(S2, ELL, XU2) + 1e-12 * np.eye(M)
    Kinv = O.cholinv(Kuu)
    pts, w = O.srcubature(m_in, P_in)
    Psi0, Psi1, Psi2 = O.psi_statistics(XU2, pts, w, S2, ELL)
    return dict(D=D, M=M, mu_y=mu_y, Sigma_y=Sigma_y, mu_v=mu_v, Sigma_v=Sigma_v, W=W, Kinv=Kinv,
                pts=pts, w=w, Psi0=Psi0, Psi1=Psi1, Psi2=Psi2, Rv=Sigma_v + np.outer(mu_v, mu_v))


def test_multi_rule_out(multi):
    """GPtest.jl:385-403: mean = kron(C, Psi1) mu_v."""
    got = O.multi_rule_out(multi["Psi1"], multi["mu_v"], multi["D"])
    gt = np.kron(np.eye(2), multi["Psi1"][None, :]) @ multi["mu_v"]
    np.testing.assert_allclose(got, gt, rtol=1e-13)


def test_multi_rule_v(multi):
    """GPtest.jl:432-456: precision kron(W, Psi2), weighted mean kron(C,Psi1)' W mu_y."""
    xi, Lam = O.multi_rule_v(multi["Psi1"], multi["Psi2"], multi["mu_y"], multi["W"])
    Psi1_tilde = np.kron(np.eye(2), multi["Psi1"][None, :])
    np.testing.assert_allclose(Lam, np.kron(multi["W"], multi["Psi2"]), rtol=1e-14)
    np.testing.assert_allclose(xi, Psi1_tilde.T @ multi["W"] @ multi["mu_y"], rtol=1e-13)


def test_multi_rule_in_closure(multi):
    """GPtest.jl:406-413: logpdf(nu_in, x) == -1/2 tr(W kron(C, A_x)) + mu_y' W kron(C, B_x) mu_v
    - 1/2 tr(Rv kron(C, B_x)' W kron(C, B_x)) at the reference's two probe inputs (and a third with a full W)."""
    C = np.eye(2)
    for W in (multi["W"], np.array([[3.0, 0.7], [0.7, 1.5]])):
        f = O.multi_rule_in_logpdf(XU2, S2, ELL, multi["mu_y"], multi["mu_v"], multi["Sigma_v"], W, multi["Kinv"])
        for x in ([1.0, 1.5], [-1.5, 2.0], [2.2, 3.1]):
            xx = np.array([x])
            B = O.kernelmatrix(S2, ELL, xx, XU2)                                            # B_x (1 x M), GPtest.jl:48
            A = O.kernelmatrix(S2, ELL, xx, xx) - B @ multi["Kinv"] @ B.T                   # A_x, GPtest.jl:47
            kB = np.kron(C, B)
            gt = (-0.5 * np.trace(W @ np.kron(C, A)) + multi["mu_y"] @ W @ kB @ multi["mu_v"]
                  - 0.5 * np.trace(multi["Rv"] @ kB.T @ W @ kB))
            assert np.isclose(f(x), gt, rtol=1e-10, atol=1e-9), (x, f(x), gt)


def test_multi_rule_theta_closure(multi):
    """GPtest.jl:473-488: the :theta closure at the reference's probes, against its trace form
    -1/2 tr(W) (Psi0 - tr(Kuu^-1 Psi2')) + Psi1 . s - 1/2 sum(Psi2' .* S)  (what the device evaluates point by point)."""
    W, M = multi["W"], multi["M"]
    kern = lambda th: (float(th[0]), np.asarray(th[1:], dtype=np.float64))                      # GPtest.jl:21
    f = O.multi_rule_theta_logpdf(XU2, kern, multi["pts"], multi["w"], multi["mu_y"], multi["mu_v"], multi["Sigma_v"], W)
    row = multi["mu_y"] @ W
    s_vec = sum(multi["mu_v"][d * M:(d + 1) * M] * row[d] for d in range(2))
    S = sum(multi["Rv"][i * M:(i + 1) * M, j * M:(j + 1) * M] * W[i, j] for i in range(2) for j in range(2))
    for th in ([1.2, 2.3], [0.5, 1.4]):
        s2, ell = kern(np.array(th))
        P0, P1, P2 = O.psi_statistics(XU2, multi["pts"], multi["w"], s2, ell)
        P2 = P2 + 1e-7 * np.eye(M)
        Kinv = O.cholinv(O.kernelmatrix(s2, ell, XU2))
        want = -0.5 * np.trace(W) * (P0 - np.trace(Kinv @ P2)) + P1 @ s_vec - 0.5 * np.sum(P2 * S)
        assert np.isclose(f(np.array(th)), want, rtol=1e-9, atol=1e-6), (th, f(np.array(th)), want)


def test_multi_rule_w(multi):
    """GPtest.jl:459-471: Wishart(D+2, inv(I1 + I2)) with Psi4 = E[kron(C,k') Rv kron(C,k)]."""
    S = O.multi_rule_w(multi["Psi0"], multi["Psi1"], multi["Psi2"], multi["mu_y"], multi["Sigma_y"],
                       multi["mu_v"], multi["Sigma_v"], multi["Kinv"])
    C = np.eye(2)
    Psi4 = np.zeros((2, 2))
    for p, wt in zip(multi["pts"], multi["w"]):
        k = O.kernelmatrix(S2, ELL, p[None, :], XU2)          # (1, M)
        Psi4 += wt * np.kron(C, k) @ multi["Rv"] @ np.kron(C, k.T)
    Psi1_tilde = np.kron(C, multi["Psi1"][None, :])
    I1 = np.kron(C, multi["Psi0"] - np.trace(multi["Kinv"] @ multi["Psi2"]))
    mu_y, mu_v = multi["mu_y"], multi["mu_v"]
    I2 = (np.outer(mu_y, mu_y) + multi["Sigma_y"] - np.outer(mu_y, mu_v) @ Psi1_tilde.T
          - Psi1_tilde @ np.outer(mu_v, mu_y) + Psi4)
    np.testing.assert_allclose(S, I1 + I2, rtol=1e-10, atol=1e-10)


def test_multi_rule_in_logpdf(multi):
    """GPtest.jl:407-413."""
    C = np.eye(2)
    for x in (np.array([1.0, 1.5]), np.array([-1.5, 2.0])):
        got = O.multi_log_backward_in(x, XU2, S2, ELL, multi["mu_y"], multi["mu_v"], multi["Sigma_v"],
                                      multi["W"], multi["Kinv"])
        k = O.kernelmatrix(S2, ELL, x[None, :], XU2)
        A = S2 - k @ multi["Kinv"] @ k.T
        B = np.kron(C, k)
        gt = (-0.5 * np.trace(multi["W"] @ np.kron(C, A)) + multi["mu_y"] @ multi["W"] @ B @ multi["mu_v"]
              - 0.5 * np.trace(multi["Rv"] @ B.T @ multi["W"] @ B))
        assert math.isclose(got, float(gt), rel_tol=1e-9)


def test_multi_average_energy(multi):
    """GPtest.jl:509-538: U = 0.5 tr(W (I1+I2)) + D/2 log 2pi - 0.5 E[logdet W]."""
    E_logdetW = 1.2345
    U = O.multi_average_energy(multi["Psi0"], multi["Psi1"], multi["Psi2"], multi["mu_y"], multi["Sigma_y"],
                               multi["mu_v"], multi["Sigma_v"], multi["W"], E_logdetW, multi["Kinv"])
    S = O.multi_rule_w(multi["Psi0"], multi["Psi1"], multi["Psi2"], multi["mu_y"], multi["Sigma_y"],
                       multi["mu_v"], multi["Sigma_v"], multi["Kinv"])
    U_gt = 0.5 * np.trace(multi["W"] @ S) + math.log(2 * math.pi) - 0.5 * E_logdetW
    assert math.isclose(U, U_gt, rel_tol=1e-10)


def test_multi_batched_equals_per_step():
    """Summed MultiSGP statistics == per-step messages folded (eq. M)."""
    rng = np.random.default_rng(5)
    T, D, M = 9, 2, 12
    Xu = rng.uniform(-2, 2, (M, 2))
    s2, ell = 0.9, np.array([1.1])
    Y = rng.normal(size=(T, D))
    Aw = rng.normal(size=(D, D))
    W = Aw @ Aw.T + np.eye(D)
    means = rng.normal(size=(T, 2))
    covs = [np.diag(rng.uniform(0.05, 0.3, 2)) for _ in range(T)]
    cub = [O.srcubature(means[t], covs[t]) for t in range(T)]
    pts = np.stack([c[0] for c in cub])
    wts = np.stack([c[1] for c in cub])
    Lam0 = np.eye(D * M) / 10.0
    xi0 = rng.normal(size=D * M) * 0.01

    Lam, xi = Lam0.copy(), xi0.copy()
    for t in range(T):
        P0, P1, P2 = O.psi_statistics(Xu, pts[t], wts[t], s2, ell)
        x_t, L_t = O.multi_rule_v(P1, P2, Y[t], W)
        Lam += L_t
        xi += x_t
    Sig_pp = O.cholinv(Lam)
    mu_pp = Sig_pp @ xi

    ms = O.multi_suff_stats(Xu, pts, wts, Y, None, s2, ell)
    mu_b, Sig_b = O.multi_v_update(ms, W, Lam0, xi0)
    np.testing.assert_allclose(mu_b, mu_pp, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(Sig_b, Sig_pp, rtol=1e-9, atol=1e-12)

    Kinv = O.cholinv(O.kernelmatrix(s2, ell, Xu) + 1e-10 * np.eye(M))
    S_sum = np.zeros((D, D))
    for t in range(T):
        P0, P1, P2 = O.psi_statistics(Xu, pts[t], wts[t], s2, ell)
        S_sum += O.multi_rule_w(P0, P1, P2, Y[t], None, mu_b, Sig_b, Kinv)
    np.testing.assert_allclose(O.multi_w_update(ms, mu_b, Sig_b, Kinv), S_sum, rtol=1e-9, atol=1e-10)
