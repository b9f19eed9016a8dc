"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Tolerance: BASELINE.json's north star asks for <= 1e-5 relative Frobenius on the
posterior; FP64 end to end lands orders of magnitude below that, the asserts use 1e-8 where the
conditioning allows and state the looser bound where it does not.
"""
import math
import os
import zlib

import numpy as np
import pytest

from oracle import sgp_oracle as O

# (The experiment lost -- DESIGN.md section 8 -- and lives in a variant library that the default build no longer produces: these
# tests run only with SGP_TEST_CHAIN=1, which also builds it.)
CHAIN_TESTS = os.environ.get("SGP_TEST_CHAIN") == "1"
needs_chain = pytest.mark.skipif(not CHAIN_TESTS, reason="persistent-chain experiment: set SGP_TEST_CHAIN=1 (builds the variant library)")

pytestmark = pytest.mark.gpu


def relF(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def kuu_tol(cond_K):
    """Bound on |L_dev - L_ref|_F / |L_ref|_F for the K_uu factor."""
    return max(1e-13, 0.05 * np.finfo(float).eps * cond_K)


def post_tol(cond_L):
    """The north star's bound is 1e-5 relative Frobenius; what FP64 can deliver is ~cond(Lambda) * eps, so the tighter."""
    return min(1e-5, max(1e-9, 20 * np.finfo(float).eps * cond_L))


@pytest.fixture(scope="module")
def G():
    import gaussianprocessnode_amd as g
    return g


def synth(N, M, D, seed, classification=False):
    """Synthetic inputs of SURVEY.md §8(d): X ~ U(-1.745, 1.745), Xu = rows of X, y = sin(sum x) + noise, standardised."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1.745, 1.745, (N, D))
    pool = X if N >= M else rng.uniform(-1.745, 1.745, (M, D))
    Xu = pool[rng.permutation(len(pool))[:M]].copy()
    y = np.sin(X.sum(axis=1)) + 0.1 * rng.normal(size=N)
    y = (y - y.mean()) / (y.std() if N > 1 else 1.0)
    vy = rng.uniform(0.05, 0.5, N) if classification else None
    return X, Xu, y, vy


# ------------------------------------------------------------------------------------------------
def test_kernelmatrix_matches_oracle(G):
    rng = np.random.default_rng(0)
    for na, nb, D in [(1, 1, 1), (7, 13, 1), (100, 33, 2), (257, 64, 8)]:
        A, B = rng.normal(size=(na, D)), rng.normal(size=(nb, D))
        ell = rng.uniform(0.5, 3.0, D)
        K = G.kernelmatrix(A, B, 0.37, ell)
        np.testing.assert_allclose(K, O.kernelmatrix(0.37, ell, A, B), rtol=1e-13, atol=1e-300)
    K1 = G.kernelmatrix(A, B, 2.0, [1.7])          # isotropic lengthscale
    np.testing.assert_allclose(K1, O.kernelmatrix(2.0, 1.7, A, B), rtol=1e-13)


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 100, 128, 200, 257, 600])
def test_potrf_potri(G, n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    A = A @ A.T / n + np.eye(n)
    L = G.potrf(A)
    assert np.allclose(np.triu(L, 1), 0.0)
    np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-12)
    Ai = G.potri(A)
    assert relF(Ai, np.linalg.inv(A)) < 1e-11
    np.testing.assert_allclose(Ai, Ai.T, rtol=0, atol=1e-13)


def _cholesky_longdouble(A):
    A = A.astype(np.longdouble)
    n = len(A)
    L = np.zeros_like(A)
    for j in range(n):
        L[j, j] = np.sqrt(A[j, j] - (L[j, :j] ** 2).sum())
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - (L[i, :j] * L[j, :j]).sum()) / L[j, j]
    return L.astype(np.float64)


@pytest.mark.parametrize("seed", [4, 5, 13, 51, 54, 58])
def test_potrf_ill_conditioned_kuu_stays_backward_stable(G, seed):
    """K_uu of the toy configuration (1-D inputs, jitter 1e-8: cond ~ 1e9).  Its last 16 x 16 block is a Schur complement of
    1e-8-sized entries -- the place where a factorisation that lets the two triangles of the block drift apart (an LU whose
    row part is thrown away) loses five digits: such a variant measured 0.12 cond eps here, the Cholesky recurrence 0.002."""
    _, Xu, _, _ = synth(50, 20, 1, seed=seed)
    K = O.kernelmatrix(0.9, np.array([1.5]), Xu) + 1e-8 * np.eye(20)
    L = G.potrf(K)
    Lr = _cholesky_longdouble(K)
    bound = np.linalg.cond(K) * np.finfo(float).eps
    assert relF(L, Lr) < 0.01 * bound, (relF(L, Lr) / bound)
    assert np.abs(L[16:, 16:] - Lr[16:, 16:]).max() < 1e-10


def test_potrf_reports_failing_minor(G):
    A = np.eye(100)
    A[70, 70] = -1.0
    with pytest.raises(G.PosDefException) as ei:
        G.potrf(A)
    assert ei.value.info == 71


# ------------------------------------------------------------------------------------------------
CASES = [
    # (name, N, M, D, w_bar, jitter, classification)
    ("toy-C1", 50, 20, 1, 100.0, 1e-8, False),          # BASELINE config 1 (GPT_regression toy)
    ("one-point", 1, 4, 1, 1.0, 1e-8, True),            # smallest graph
    ("ragged", 333, 37, 3, 10.0, 1e-8, False),          # nothing a multiple of anything
    ("banana-C4", 1000, 128, 2, 3.0, 1e-8, True),       # config 4 shape, classification (v_y present)
    ("kin40k-C2", 2000, 256, 8, 1e4, 0.0, False),       # config 2 shape, reduced N
    ("kin40k-C2-full", 10000, 256, 8, 1e4, 0.0, False), # BASELINE config 2 at its full size
    ("kin40k-T", 1500, 512, 8, 1e4, 0.0, False),        # north-star M
    ("kin40k-M600", 500, 600, 8, 1e4, 0.0, False),      # the reference's real M with one minibatch (N < M)
]


@pytest.mark.parametrize("name,N,M,D,w,jit,cls", CASES, ids=[c[0] for c in CASES])
def test_sweep_matches_oracle(G, name, N, M, D, w, jit, cls):
    X, Xu, y, vy = synth(N, M, D, seed=zlib.crc32(name.encode()) % 1000, classification=cls)
    s2 = 0.9
    ell = np.linspace(1.5, 3.0, D)
    E_logw = math.log(w) - 0.01
    with G.SGPDevice(N, M, D, keep_kuf=True) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y, vy)
        dev.set_kernel(s2, ell, jit)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]], E_logw)
        dev.sweep()
        Psi2, B, sc_data = dev.stats()
        KuuL = dev.kuu_chol()
        mu, Sig, Uv = dev.posterior()
        sc = dev.scalars()
        I1, I2 = dev.w_stats()
        obj = dev.theta_objective()
    ref = O.vmp_sweep(Xu, X, y, vy, s2, ell, w, E_logw=E_logw, jitter=jit, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    # statistics (the all-reduce payload)
    assert relF(Psi2, ref.stats.Psi2) < 1e-13
    assert relF(B, ref.stats.b) < 1e-13
    assert math.isclose(sc_data[0], ref.stats.s_yy[0, 0], rel_tol=1e-13)
    assert sc_data[1] == N and sc_data[2] == N
    # K_uu factor: two FP64 Cholesky factorisations differ by a small multiple of cond(K_uu) * eps (the toy shape has
    # cond ~ 1e9-1e10; the worst ratio over 300 random draws is 0.006, profiles/r01_accuracy_sweep.txt)
    Kuu = O.kernelmatrix(s2, ell, Xu) + jit * np.eye(M)
    cond_K = np.linalg.cond(Kuu)
    assert relF(KuuL, ref.KuuL) < kuu_tol(cond_K), (relF(KuuL, ref.KuuL), cond_K)
    # posterior: the north star's bound is 1e-5 relative Frobenius; what FP64 can deliver is ~cond(Lambda) * eps
    # (two independent FP64 evaluations differ by that much), so assert the tighter of the two.
    cond_L = np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2)
    tol_post = post_tol(cond_L)
    assert relF(mu, ref.mu_v) < tol_post, (relF(mu, ref.mu_v), cond_L)
    assert relF(Sig, ref.Sigma_v) < tol_post, (relF(Sig, ref.Sigma_v), cond_L)
    assert relF(Uv, ref.Uv) < tol_post
    assert np.allclose(np.tril(Uv, -1), 0.0)
    # summed :w messages and average energy.  sum I1 = s_kk - tr(Kuu^-1 Psi2) cancels against s_kk, so its
    # attainable accuracy is cond(Kuu) * eps * s_kk (see test_oracle_identities); the toy case has cond ~ 1e9.
    tol_I1 = 50 * np.finfo(float).eps * cond_K * ref.stats.s_kk + 1e-12
    assert abs(sc.sum_I1 - ref.sum_I1) <= tol_I1
    assert math.isclose(sc.sum_I2, ref.sum_I2, rel_tol=max(1e-7, tol_post))
    assert abs(sc.energy - ref.energy) <= max(1e-7, tol_post) * abs(ref.energy) + 0.5 * w * tol_I1
    assert sc.info_kuu == 0 and sc.info_lambda == 0
    assert math.isclose(sc.logdet_kuu, 2 * np.log(np.diag(ref.KuuL)).sum(), rel_tol=1e-9, abs_tol=1e-7)
    # per-point :w quantities (Q_ff diagonal term)
    rI1, rI2 = O.w_stats_perpoint(Xu, X, y, vy, s2, ell, ref.KuuL, ref.mu_v, ref.Uv)
    np.testing.assert_allclose(I1, rI1, rtol=0, atol=tol_I1 / N + 1e-12)
    # I2_n = y^2 + v - 2 y k.mu + |Uv k|^2 cancels too: absolute accuracy = posterior accuracy x the terms' size
    scale_I2 = float(np.max(y * y + np.sum((ref.Uv @ O.kernelmatrix(s2, ell, Xu, X)) ** 2, axis=0)))
    np.testing.assert_allclose(I2, rI2, rtol=1e-6, atol=max(1e-9, tol_post * scale_I2))
    # theta objective at the sweep's own posterior (helper_functions/derivative_helper.jl:23-39)
    ref_obj = O.theta_objective(Xu, X, y, s2, ell, ref.mu_v, ref.Uv, w, jitter=jit)
    assert abs(obj - ref_obj) <= 1e-7 * abs(ref_obj) + 0.5 * w * tol_I1


def test_sweep_with_more_tiles_than_one_round_of_slabs(G):
    """M = 3000: 47 tile rows = 1128 lower tiles, more than the 4 x 256 workgroups of one SYRK round -- the slab area must hold at
    least one slab per tile (a fixed 1024-slab capacity refused every M above 2816).  Statistics and posterior
    against the oracle (GPnode/UniSGPnode.jl:62-73,144-173 batched)."""
    N, M, D, w, jit = 300, 3000, 8, 50.0, 1e-6
    X, Xu, y, _ = synth(N, M, D, seed=77)
    s2, ell = 0.8, np.full(D, 0.9)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(s2, ell, jit)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        dev.sweep()
        Psi2, B, _ = dev.stats()
        mu, Sig, Uv = dev.posterior()
        sc = dev.scalars()
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=jit, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    assert relF(Psi2, ref.stats.Psi2) < 1e-13 and relF(B, ref.stats.b) < 1e-13
    tol = post_tol(np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2))
    assert relF(mu, ref.mu_v) < tol and relF(Sig, ref.Sigma_v) < tol and relF(Uv, ref.Uv) < tol
    assert sc.info_kuu == 0 and sc.info_lambda == 0
    assert math.isclose(sc.logdet_kuu, 2 * np.log(np.diag(ref.KuuL)).sum(), rel_tol=1e-9, abs_tol=1e-6)


SWEEP_SHAPES = [("toy", 50, 20, 1, 100.0, 1e-8), ("ragged", 333, 37, 3, 10.0, 1e-8), ("mid", 700, 130, 2, 30.0, 1e-8)]


@pytest.mark.parametrize("name,N,M,D,w,jit", SWEEP_SHAPES, ids=[c[0] for c in SWEEP_SHAPES])
def test_accuracy_over_random_draws(G, name, N, M, D, w, jit):
    """tests/scripts/accuracy_sweep.py as a test: 24 random draws per shape (the toy shape has cond(K_uu) ~ 1e9-1e10), K_uu factor
    and posterior against the oracle at the conditioning-aware bounds of test_sweep_matches_oracle.  A variant of the pivot
    loop that let the two triangles of the diagonal block drift apart passed the fixed-seed cases and failed 8 of 120 draws
    of this kind; the fixed-seed cases alone do not guard the factorisation kernels."""
    eps = np.finfo(float).eps
    worst = {"kuu": 0.0, "mu": 0.0, "sig": 0.0, "uv": 0.0}
    s2, ell = 0.9, np.linspace(1.5, 3.0, D)
    with G.SGPDevice(N, M, D) as dev:
        for seed in range(24):
            X, Xu, y, vy = synth(N, M, D, seed=1000 + seed)
            dev.set_inducing(Xu)
            dev.set_data(X, y, vy)
            dev.set_kernel(s2, ell, jit)
            dev.set_prior_isotropic(50.0)
            dev.set_noise([[w]], math.log(w) - 0.01)
            dev.sweep()
            KuuL = dev.kuu_chol()
            mu, Sig, Uv = dev.posterior()
            ref = O.vmp_sweep(Xu, X, y, vy, s2, ell, w, E_logw=math.log(w) - 0.01, jitter=jit, Lambda0=np.eye(M) / 50.0,
                              xi0=np.zeros(M))
            cK = np.linalg.cond(O.kernelmatrix(s2, ell, Xu) + jit * np.eye(M))
            cL = np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2)
            errs = {"kuu": relF(KuuL, ref.KuuL), "mu": relF(mu, ref.mu_v), "sig": relF(Sig, ref.Sigma_v), "uv": relF(Uv, ref.Uv)}
            assert errs["kuu"] < kuu_tol(cK), (seed, errs, cK)
            for k in ("mu", "sig", "uv"):
                assert errs[k] < post_tol(cL), (seed, k, errs, cL)
            worst["kuu"] = max(worst["kuu"], errs["kuu"] / (cK * eps))
            for k in ("mu", "sig", "uv"):
                worst[k] = max(worst[k], errs[k] / (cL * eps))
    print(name, "worst error / (cond * eps):", {k: round(v, 4) for k, v in worst.items()})


def test_prior_forms_agree(G):
    N, M, D = 400, 48, 2
    X, Xu, y, _ = synth(N, M, D, seed=5)
    rng = np.random.default_rng(5)
    A = rng.normal(size=(M, M))
    Sigma0 = A @ A.T / M + 0.5 * np.eye(M)
    mu0 = rng.normal(size=M) * 0.3
    Lam0 = np.linalg.inv(Sigma0)
    s2, ell, w = 1.1, np.array([1.0, 2.0]), 25.0
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-8, mu0=mu0, Sigma0=Sigma0)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(s2, ell, 1e-8)
        dev.set_noise([[w]])
        dev.set_prior_meancov(mu0, Sigma0)
        dev.sweep()
        mu_a, Sig_a, _ = dev.posterior()
        dev.set_prior_precision(Lam0 @ mu0, Lam0)
        dev.sweep()
        mu_b, Sig_b, _ = dev.posterior()
    for mu, Sig in ((mu_a, Sig_a), (mu_b, Sig_b)):
        assert relF(mu, ref.mu_v) < 1e-8
        assert relF(Sig, ref.Sigma_v) < 1e-8


def test_minibatch_carry_matches_full_batch(G):
    """posterior -> prior carry over minibatches (experiments/regression_kin40k.ipynb:203-212), ragged last batch."""
    N, M, D = 1100, 64, 4
    X, Xu, y, _ = synth(N, M, D, seed=9)
    s2, ell, w = 0.7, np.full(D, 2.0), 1e3
    full = O.vmp_sweep(Xu, X, y, None, s2, ell, w, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    xb, yb = O.split2batch(X, y, 500)
    with G.SGPDevice(500, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_kernel(s2, ell, 0.0)
        dev.set_noise([[w]])
        mu, Sig = np.zeros(M), 50.0 * np.eye(M)
        for xi, yi in zip(xb, yb):
            dev.set_prior_meancov(mu, Sig)
            dev.set_data(xi, yi)
            dev.sweep()
            mu, Sig, _ = dev.posterior(want_uv=False)
        # the same carry kept on the device, in natural form (sgp_carry_posterior)
        dev.set_prior_isotropic(50.0)
        for xi, yi in zip(xb, yb):
            dev.set_data(xi, yi)
            dev.sweep()
            dev.carry_posterior()
        mu_d, Sig_d, _ = dev.posterior(want_uv=False)
        dev.theta_objective()                          # same theta as the sweep: nothing recomputed, carry still valid
        dev.carry_posterior()
        dev.set_kernel(s2 * 1.1, ell, 0.0)
        dev.theta_objective()                          # new theta: statistics re-evaluated ...
        with pytest.raises(G.SGPError):
            dev.carry_posterior()                      # ... and no longer belong to q(v)'s sweep
    assert relF(mu, full.mu_v) < 1e-7
    assert relF(Sig, full.Sigma_v) < 1e-7
    assert relF(mu_d, full.mu_v) < 1e-7                  # conditioning-limited like the host carry above
    assert relF(Sig_d, full.Sigma_v) < 1e-7


def test_graph_replay_equals_eager_and_tracks_parameters(G):
    N, M, D = 700, 96, 3
    X, Xu, y, _ = synth(N, M, D, seed=3)
    outs = []
    # graph replay against eager launches of the same kernels: bitwise; against the opt-in persistent factorisation launch
    # (SGP_FLAG_PERSISTENT_CHAIN, right-looking): to rounding
    variants = ((True, False), (False, False), (False, True)) if CHAIN_TESTS else ((True, False), (False, False), (False, False))
    for use_graph, persistent in variants:
        with G.SGPDevice(N, M, D, use_graph=use_graph, persistent_chain=persistent) as dev:
            dev.set_inducing(Xu)
            dev.set_data(X, y)
            dev.set_prior_isotropic(50.0)
            res = []
            for s2, w in ((1.0, 10.0), (0.5, 200.0), (1.0, 10.0)):      # parameters change between replays
                dev.set_kernel(s2, np.array([1.0, 1.5, 2.0]), 1e-8)
                dev.set_noise([[w]])
                dev.sweep()
                mu, Sig, Uv = dev.posterior()
                res.append((mu, Sig, Uv, dev.scalars().energy))
            outs.append(res)
    for a, b, c in zip(*outs):
        for u, v, x in zip(a[:3], b[:3], c[:3]):
            assert np.array_equal(u, v)          # same kernels, same order: bitwise equal
            assert relF(x, v) < 1e-9             # the persistent launch factors right-looking: equal to ~cond * eps
        assert a[3] == b[3]
        assert math.isclose(c[3], b[3], rel_tol=1e-6)      # (the energy cancels against s_kk: cond(K_uu) * eps)
    assert np.array_equal(outs[0][0][0], outs[0][2][0])      # same parameters again -> same result
    assert not np.array_equal(outs[0][0][0], outs[0][1][0])


def test_resident_parameters_are_refreshed_by_every_setter(G):
    """Sweeps at unchanged parameters skip the parameter/inducing-input mirror kernel: every setter (kernel, noise, prior,
    inducing inputs), a prediction and a theta-objective evaluation at other parameters in between must still leave the next
    sweep bitwise equal to the same sweep on a fresh handle."""
    N, M, D = 600, 70, 3
    X, Xu, y, _ = synth(N, M, D, seed=8)
    Xu2 = Xu + 0.05
    ell_a, ell_b = np.array([1.0, 1.5, 2.0]), np.array([0.8, 1.1, 2.5])

    def fresh(Xu_, s2, ell, w, pv):
        with G.SGPDevice(N, M, D) as d:
            d.set_inducing(Xu_); d.set_data(X, y); d.set_kernel(s2, ell, 1e-8); d.set_prior_isotropic(pv); d.set_noise([[w]])
            d.sweep()
            return d.posterior() + (d.scalars().energy,)

    def same(a, b):
        return all(np.array_equal(u, v) for u, v in zip(a[:3], b[:3])) and a[3] == b[3]

    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, ell_a, 1e-8); dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
        dev.sweep(); dev.sweep(); dev.sweep()                                    # the 2nd and 3rd start with the Gram kernel
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu, 1.0, ell_a, 10.0, 50.0))
        dev.set_kernel(0.7, ell_b, 1e-8); dev.sweep(); dev.sweep()
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu, 0.7, ell_b, 10.0, 50.0))
        dev.set_noise([[33.0]]); dev.sweep()
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu, 0.7, ell_b, 33.0, 50.0))
        dev.set_prior_isotropic(5.0); dev.sweep()
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu, 0.7, ell_b, 33.0, 5.0))
        dev.set_inducing(Xu2); dev.sweep(); dev.sweep()
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu2, 0.7, ell_b, 33.0, 5.0))
        mu = dev.posterior()[0]
        dev.predict(X[:50], mu)                                                  # mirrors the parameters itself
        dev.set_kernel(1.3, ell_a, 1e-8)
        dev.theta_objective(want_grad=True)                                      # evaluated at the new theta, q(v) of the old sweep
        dev.set_kernel(0.7, ell_b, 1e-8); dev.sweep()
        assert same(dev.posterior() + (dev.scalars().energy,), fresh(Xu2, 0.7, ell_b, 33.0, 5.0))
        tot, cnt = dev.phase_totals()
        assert cnt >= 9 and all(t >= 0 for t in tot)                             # the phase stamps survived the skipped kernels


def test_phase_stamps_of_the_last_sweep(G):
    """sgp_get_timestamps: (begin, end) of every phase of the LAST sweep on the 100 MHz clock, kept by the sweep's closing
    kernel; the sweep record spans the phases and agrees with the accumulated totals."""
    N, M, D = 3000, 256, 4
    X, Xu, y, _ = synth(N, M, D, seed=12)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-8); dev.set_prior_isotropic(50.0)
        dev.set_noise([[10.0]])
        for _ in range(3):
            dev.sweep()
        dev.phase_totals(reset=True)
        dev.sweep()                                                     # a sweep that starts with the Gram kernel
        ts = np.asarray(dev.timestamps(), dtype=np.int64).reshape(-1, 2)
        tot, cnt = dev.phase_totals()
    assert cnt == 1
    SWEEP, GRAM, SYRK, FINISH1, FINISH2, LOCAL = 0, 1, 2, 3, 4, 7
    for slot in (SWEEP, GRAM, SYRK, FINISH1, FINISH2, LOCAL):
        b, e = ts[slot]
        assert 0 < b < e < b + 100 * 1_000_000, (slot, b, e)            # set, ordered, shorter than a second
    assert ts[SWEEP][0] <= ts[GRAM][0] and ts[LOCAL][1] <= ts[FINISH1][1] <= ts[SWEEP][1]
    assert ts[GRAM][1] <= ts[SYRK][1] <= ts[LOCAL][1]
    assert abs((ts[SWEEP][1] - ts[SWEEP][0]) / 100.0 - tot[SWEEP]) < 0.02   # the same sweep, in microseconds


def test_two_phase_with_bound_statistics_buffer(G):
    """The multi-GPU hand-off on one GPU: two shards' statistics summed in a caller-owned buffer
    (what the RCCL all-reduce does), then the replicated finish."""
    torch = pytest.importorskip("torch")
    N, M, D = 900, 80, 2
    X, Xu, y, _ = synth(N, M, D, seed=21)
    s2, ell, w = 1.0, np.array([1.3, 0.8]), 50.0
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-8, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    cut = 400
    devs = []
    for sl in (slice(0, cut), slice(cut, N)):
        d = G.SGPDevice(N, M, D)
        d.set_inducing(Xu)
        d.set_data(X[sl], y[sl])
        d.set_kernel(s2, ell, 1e-8)
        d.set_prior_isotropic(50.0)
        d.set_noise([[w]])
        devs.append(d)
    _, count, _ = devs[0].stats_layout()
    bufs = [torch.zeros(count, dtype=torch.float64, device="cuda") for _ in devs]
    tstream = torch.cuda.Stream()                 # an explicit stream: a NULL stream means "the library's own" to the ABI
    torch.cuda.synchronize()
    with torch.cuda.stream(tstream):
        for d, b in zip(devs, bufs):
            d.bind_stats(b.data_ptr())
            d.sweep_local(tstream.cuda_stream)
        total = bufs[0] + bufs[1]                 # stands in for the all-reduce, ordered on the same stream
        for d, b in zip(devs, bufs):
            b.copy_(total)
            d.sweep_finish(tstream.cuda_stream)
    torch.cuda.synchronize()
    for d in devs:
        mu, Sig, _ = d.posterior(want_uv=False)
        sc = d.scalars()
        assert relF(mu, ref.mu_v) < 1e-8 and relF(Sig, ref.Sigma_v) < 1e-8
        assert math.isclose(sc.sum_I2, ref.sum_I2, rel_tol=1e-8)
        d.close()


def test_lambda_not_positive_definite_is_reported(G):
    N, M, D = 100, 16, 1
    X, Xu, y, _ = synth(N, M, D, seed=1)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(1.0, [1.0], 1e-8)
        dev.set_noise([[-10.0]], 0.0)            # negative precision -> Lambda indefinite
        dev.set_prior_isotropic(50.0)
        dev.sweep()
        with pytest.raises(G.PosDefException):
            dev.posterior()


def test_argument_errors(G):
    with pytest.raises(G.SGPError):
        G.SGPDevice(10, 0, 1)
    with G.SGPDevice(10, 4, 2) as dev:
        with pytest.raises(G.SGPError):
            dev.sweep()                           # nothing set yet
        with pytest.raises(G.SGPError):
            dev.set_data(np.zeros((11, 2)), np.zeros(11))      # n > n_max
        with pytest.raises(G.SGPError):
            dev.set_kernel(1.0, [1.0, 2.0, 3.0])               # wrong number of lengthscales


# ------------------------------------------------------------------------------------------------
def test_predict_matches_oracle_and_golden_kin40k(G, golden):
    fx = golden("kin40k_fixture")
    data = golden("kin40k_data")
    s2, ell = O.kernel_from_theta(fx["theta_opt"], softplus_params=True)
    M, D = fx["Xu"].shape
    with G.SGPDevice(16, M, D) as dev:
        dev.set_inducing(fx["Xu"])
        dev.set_kernel(s2, ell, 1e-8)
        pred = dev.predict(data["xtest"], fx["mu_v"])
    ref = O.predict_mean(fx["Xu"], data["xtest"], fx["mu_v"], s2, ell)
    assert relF(pred, ref) < 1e-12
    assert abs(O.SMSE(data["ytest"], pred) - 0.08343114079545057) < 1e-9      # experiments/regression_kin40k.ipynb:315


def test_banana_errors_golden(G, golden):
    fx = golden("banana_fixture")
    s2, ell = O.kernel_from_theta(fx["theta_opt"], softplus_params=True)
    xtest, ytest = fx["data"][4000:, :2], (fx["data"][4000:, 2] + 1) / 2
    with G.SGPDevice(16, 500, 2) as dev:
        dev.set_inducing(fx["Xu"])
        dev.set_kernel(s2, ell, 1e-8)
        f = dev.predict(xtest, fx["mu_v"])
    assert O.num_error(ytest, (f >= 0).astype(float)) == 125.0              # experiments/classification_banana.ipynb:316


def test_kin40k_full_training_sweep_real_data(G, golden):
    """BASELINE config 2/T on the real kin40k training set with the reference's own Xu / theta_opt (M = 600)."""
    fx = golden("kin40k_fixture")
    data = golden("kin40k_data")
    s2, ell = O.kernel_from_theta(fx["theta_opt"], softplus_params=True)
    M, D = fx["Xu"].shape
    with G.SGPDevice(10000, M, D) as dev:
        dev.set_inducing(fx["Xu"])
        dev.set_data(data["xtrain"], data["ytrain"])
        dev.set_kernel(s2, ell, 0.0)                  # no jitter in the training Cholesky (:183-184)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[1e4]])
        dev.sweep()
        mu, Sig, Uv = dev.posterior()
        pred = dev.predict(data["xtest"])
    ref = O.vmp_sweep(fx["Xu"], data["xtrain"], data["ytrain"], None, s2, ell, 1e4,
                      Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    assert relF(mu, ref.mu_v) < 1e-6, relF(mu, ref.mu_v)        # cond(Lambda) ~ 7e8
    assert relF(Sig, ref.Sigma_v) < 1e-6
    assert relF(Uv, ref.Uv) < 1e-6
    # and it lands where the reference's saved posterior is (theta drifted in its last epoch: loose)
    assert relF(mu, fx["mu_v"]) < 5e-3
    assert abs(O.SMSE(data["ytest"], pred) - 0.0834) < 2e-3


# ------------------------------------------------------------------------------------------------
# MultiSGP (GPnode/MultiSGPnode.jl): cubature points as weighted data, Kronecker precision, Wishart statistics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,M,Do,gauss_out", [(30, 48, 2, False), (17, 20, 3, True), (300, 48, 2, True), (7, 65, 3, False),
                                                (33, 21, 4, True), (60, 130, 2, True), (5, 1, 2, False),
                                                (1000, 70, 2, True), (601, 130, 2, False)])
def test_multisgp_sweep_matches_oracle(G, T, M, Do, gauss_out):
    """BASELINE config 5 shape (pendulum: 300 steps, M=48, D=2, srcubature = 5 points per step).  The last two cases are large
    enough (points x lower tiles >= 10 000) for the gated K_uu chain and k_syrk_direct with per-point weights -- 3 005 points: a
    ragged last k-step."""
    rng = np.random.default_rng(T + M)
    Din = 2
    Xu = rng.uniform(-2, 2, (M, Din))
    s2, ell = 0.8, np.array([1.3, 0.9])
    means = rng.normal(size=(T, Din))
    covs = [np.diag(rng.uniform(0.02, 0.2, Din)) for _ in range(T)]
    cub = [O.srcubature(means[t], covs[t]) for t in range(T)]
    pts = np.stack([c[0] for c in cub])                    # (T, S, Din)
    wts = np.stack([c[1] for c in cub])                    # (T, S)
    S = pts.shape[1]
    Y = rng.normal(size=(T, Do))
    Sig_y = np.stack([np.diag(rng.uniform(0.01, 0.1, Do)) for _ in range(T)]) if gauss_out else None
    A = rng.normal(size=(Do, Do))
    W = A @ A.T + Do * np.eye(Do)
    E_logdetW = float(np.linalg.slogdet(W)[1]) - 0.1
    Q = Do * M
    Lam0 = np.eye(Q) / 10.0
    xi0 = 0.01 * rng.normal(size=Q)

    ms = O.multi_suff_stats(Xu, pts, wts, Y, Sig_y, s2, ell)
    mu_ref, Sig_ref = O.multi_v_update(ms, W, Lam0, xi0)
    Kinv = O.cholinv(O.kernelmatrix(s2, ell, Xu) + 1e-10 * np.eye(M))
    S_ref = O.multi_w_update(ms, mu_ref, Sig_ref, Kinv)
    U_ref = 0.0
    for t in range(T):
        P0, P1, P2 = O.psi_statistics(Xu, pts[t], wts[t], s2, ell)
        U_ref += O.multi_average_energy(P0, P1, P2, Y[t], None if Sig_y is None else Sig_y[t], mu_ref, Sig_ref, W,
                                        E_logdetW, Kinv)

    with G.SGPDevice(T * S, M, Din, d_out=Do) as dev:
        dev.set_inducing(Xu)
        dev.set_data(pts.reshape(T * S, Din), np.repeat(Y, S, axis=0), None, wts.reshape(-1), n_nodes=T)
        if gauss_out:
            dev.set_output_cov_sum(Sig_y.sum(axis=0))
        dev.set_kernel(s2, ell, 1e-10)
        dev.set_prior_precision(xi0, Lam0)
        dev.set_noise(W, E_logdetW)
        dev.sweep()
        Psi2, B, sc = dev.stats()
        mu, Sig, Uv = dev.posterior()
        Sw = dev.wishart_invscale()
        energy = dev.scalars().energy
    assert relF(Psi2, ms.Psi2) < 1e-12 and relF(B, ms.B) < 1e-12
    assert sc[2] == T and math.isclose(sc[1], T, rel_tol=1e-12)
    assert relF(mu, mu_ref) < 1e-8 and relF(Sig, Sig_ref) < 1e-8
    np.testing.assert_allclose(Uv.T @ Uv, Sig_ref + np.outer(mu_ref, mu_ref), rtol=1e-7, atol=1e-10)
    # the diagonal of S carries sum I1 = s_kk - tr(Kuu^-1 Psi2), which cancels: attainable accuracy cond(Kuu) * eps * s_kk
    # (inducing inputs crowded into [-2, 2]^2 make cond(Kuu) ~ 1e12 at M = 130); the off-diagonal has no such term
    Kuu = O.kernelmatrix(s2, ell, Xu) + 1e-10 * np.eye(M)
    tol_I1 = 50 * np.finfo(float).eps * np.linalg.cond(Kuu) * s2 * T
    assert np.abs(Sw - S_ref).max() <= 1e-7 * np.abs(S_ref).max() + tol_I1, (Sw, S_ref)
    off = ~np.eye(Do, dtype=bool)
    np.testing.assert_allclose(Sw[off], S_ref[off], rtol=1e-7, atol=1e-9)
    assert abs(energy - U_ref) <= 1e-7 * abs(U_ref) + 0.5 * np.trace(W) * tol_I1, (energy, U_ref)


def test_repeated_sweeps_on_real_data_are_bitwise_identical(G, golden):
    """Regression test for an inter-block hazard (a block overwriting a tile other blocks still read): with the side
    stream competing for CUs it showed up as run-to-run differences / spurious PosDef failures.  Fixed summation order
    everywhere makes identical inputs give identical bits."""
    fx = golden("kin40k_fixture")
    data = golden("kin40k_data")
    s2, ell = O.kernel_from_theta(fx["theta_opt"], softplus_params=True)
    M, D = fx["Xu"].shape
    outs = []
    with G.SGPDevice(10000, M, D) as dev:
        dev.set_inducing(fx["Xu"])
        dev.set_data(data["xtrain"], data["ytrain"])
        dev.set_kernel(s2, ell, 0.0)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[1e4]])
        for _ in range(6):
            dev.sweep()
            mu, Sig, Uv = dev.posterior()
            outs.append((mu, Sig, Uv, dev.scalars().energy))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
        assert o[3] == outs[0][3]


@pytest.mark.parametrize("N,M,D,iso,jitter,weighted", [(600, 48, 3, False, 0.0, False), (2000, 200, 8, False, 0.0, False),
                                                      (700, 70, 2, True, 1e-6, False), (500, 64, 4, False, 1e-8, True)])
def test_theta_objective_and_gradient_at_fixed_posterior(G, N, M, D, iso, jitter, weighted):
    """neg_log_backwardmess_fast / grad_llh_new! (helper_functions/derivative_helper.jl:23-39,59-63) at a
    NEW theta with q(v) held at the last sweep -- the call pattern of experiments/regression_kin40k.ipynb:212-221.
    The device gradient is analytic; the check is central differences of the ORACLE objective (and, for cubature point
    weights, which the reference objective does not have, central differences of the device objective)."""
    rng = np.random.default_rng(N + M)
    X, Xu, y, _ = synth(N, M, D, seed=13)
    s2, w = 0.9, 200.0
    ell = np.full(D, 1.6) if iso else rng.uniform(1.0, 2.2, D)
    om = rng.uniform(0.2, 1.5, N) if weighted else None
    s2n = 1.05
    elln = np.full(D, 1.45) if iso else ell * rng.uniform(0.9, 1.1, D)      # the optimiser's next theta
    n_ell = 1 if iso else D
    p0 = np.concatenate([[s2n], elln[:n_ell]])
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y, weights=om)
        dev.set_kernel(s2, ell[:n_ell], jitter)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        dev.sweep()
        mu0, Sig0, Uv0 = dev.posterior()
        dev.set_kernel(s2n, elln[:n_ell], jitter)
        val, grad = dev.theta_objective(want_grad=True, n_ell=n_ell)
        mu1, _, _ = dev.posterior(want_cov=False, want_uv=False)

        def f_dev(p):
            dev.set_kernel(p[0], p[1:], jitter)
            return dev.theta_objective(want_grad=False, n_ell=n_ell)
        g_dev = np.array([(f_dev(p0 + 1e-5 * e) - f_dev(p0 - 1e-5 * e)) / 2e-5 for e in np.eye(1 + n_ell)])
    assert np.array_equal(mu0, mu1)                              # q(v) untouched
    # central differences of an objective that cancels against tr(Kuu^-1 Psi2): good to ~1e-5 relative when cond(Kuu) is large
    np.testing.assert_allclose(grad, g_dev, rtol=5e-5, atol=1e-6 * np.abs(g_dev).max())
    if not weighted:
        full = lambda p: p[1:] if not iso else np.full(D, p[1])
        f = lambda p: O.theta_objective(Xu, X, y, p[0], full(p), mu0, Uv0, w, jitter=jitter)
        g_ref = np.array([(f(p0 + 1e-6 * e) - f(p0 - 1e-6 * e)) / 2e-6 for e in np.eye(1 + n_ell)])
        assert math.isclose(val, f(p0), rel_tol=1e-8), (val, f(p0))
        np.testing.assert_allclose(grad, g_ref, rtol=5e-5, atol=1e-6 * np.abs(g_ref).max())


@pytest.mark.parametrize("N,weighted", [(6007, False), (6007, True), (6006, True), (12289, False)])
def test_direct_syrk_ragged_chunks_and_point_weights(G, N, weighted):
    """k_syrk_direct (the SYRK of every problem that fills the chip: N x lower tiles >= 10 000) on point counts that are no multiple of
    the 4-point k-step -- the ragged last step of a chunk is formed unpipelined by one wave -- and with per-point weights omega
    (cubature points, GPnode/UniSGPnode.jl:153-156 with Appendix A's omega): Psi2, B and the per-point :w quantities against the
    oracle; the overlapped and the plain order agree."""
    M, D = 512, 8
    rng = np.random.default_rng(N)
    X, Xu, y, _ = synth(N, M, D, seed=N % 97)
    om = rng.uniform(0.2, 1.5, N) if weighted else None
    s2, ell, w = 0.9, np.linspace(1.5, 3.0, D), 50.0
    out = {}
    for order in ("overlapped", "plain"):
        os.environ["SGP_OVERLAP"] = "1" if order == "overlapped" else "0"
        try:
            with G.SGPDevice(N, M, D, keep_kuf=True) as dev:
                dev.set_inducing(Xu); dev.set_data(X, y, weights=om); dev.set_kernel(s2, ell, 1e-8)
                dev.set_prior_isotropic(50.0); dev.set_noise([[w]])
                plan = dev.overlap_plan()
                assert bool(plan) == (order == "overlapped")
                dev.sweep()
                out[order] = (dev.stats(), dev.posterior(), dev.w_stats())
        finally:
            os.environ.pop("SGP_OVERLAP", None)
    (Psi2, B, _), (mu, Sig, Uv), (I1, I2) = out["overlapped"]
    st = O.suff_stats(Xu, X, y, None, s2, ell, omega=om)
    assert relF(Psi2, st.Psi2) < 1e-13
    assert relF(B, st.b) < 1e-13
    (Psi2p, Bp, _), (mup, _, _), (I1p, I2p) = out["plain"]
    assert relF(Psi2p, st.Psi2) < 1e-13 and relF(Bp, st.b) < 1e-13
    assert relF(mu, mup) < 1e-7             # (the two orders bracket the sums differently: cond(Lambda) x eps)
    KuuL = np.linalg.cholesky(O.kernelmatrix(s2, ell, Xu) + 1e-8 * np.eye(M))
    rI1, rI2 = O.w_stats_perpoint(Xu, X, y, None, s2, ell, KuuL, mu, Uv)
    cond_K = np.linalg.cond(KuuL) ** 2
    np.testing.assert_allclose(I1, rI1, rtol=0, atol=50 * np.finfo(float).eps * cond_K * s2 + 1e-12)
    np.testing.assert_allclose(I2, rI2, rtol=1e-6, atol=1e-8 * float(np.max(rI2)))


def _sc(dev):
    import dataclasses
    return np.array(dataclasses.astuple(dev.scalars()), dtype=float)


@pytest.mark.parametrize("N,M,D", [(10000, 512, 8), (4000, 128, 2)])
def test_one_sweep_at_a_time_is_bitwise_the_back_to_back_sweep(G, N, M, D):
    """A caller that fetches something after every sweep (sgp_get_scalars / sgp_w_stats between two sgp_sweep calls) gets its scalars
    through the pinned mirror k_scalars writes, and sgp_sweep enqueues the two chains' launches alternately:
    posterior, scalars and per-point quantities are bitwise those of the plain host order (SGP_INTERLEAVE=0) and of the copies
    (SGP_NO_ZERO_COPY=1), and the scalars equal what sgp_get_posterior's own status check and a later getter see."""
    X, Xu, y, _ = synth(N, M, D, seed=5)
    s2, ell, w = 0.9, np.linspace(1.5, 3.0, D), 50.0
    out = {}
    for mode, env in (("default", {}), ("never", {"SGP_INTERLEAVE": "0"}), ("copies", {"SGP_NO_ZERO_COPY": "1"})):
        os.environ.update(env)
        try:
            with G.SGPDevice(N, M, D, keep_kuf=True) as dev:
                dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(s2, ell, 1e-8)
                dev.set_prior_isotropic(50.0); dev.set_noise([[w]])
                sc = []
                for it in range(6):                     # one at a time
                    dev.sweep()
                    sc.append(_sc(dev) if it % 2 == 0 else np.concatenate(dev.w_stats())[:8])
                final = _sc(dev)
                again = _sc(dev)          # (a second getter without a sweep in between)
                assert np.array_equal(final, again)
                dev.sweep(); dev.sweep()                # back to back, then one fetch
                out[mode] = (sc, final, _sc(dev), dev.posterior(), dev.w_stats())
        finally:
            for k in env:
                os.environ.pop(k, None)
    ref = out["never"]
    for mode in ("default", "copies"):
        got = out[mode]
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b), mode
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), mode
        for a, b in zip(got[3], ref[3]):
            assert np.array_equal(a, b), mode
        for a, b in zip(got[4], ref[4]):
            assert np.array_equal(a, b), mode
    assert np.array_equal(ref[1], ref[2])              # (same inputs: every sweep gives the same scalars)


def test_full_size_configs_and_size_independent_properties(G):
    """BASELINE's full sizes: C3 (N = 40 000, M = 512, D = 8) against the oracle, plus properties that hold at any size:
    statistics are additive over a split and invariant under a permutation of the points; at N = 10^6 (4 GB of K_uf) the
    sweep still satisfies Uv'Uv = Sigma + mu mu' and Lambda Sigma = I."""
    N, M, D = 40000, 512, 8
    X, Xu, y, _ = synth(N, M, D, seed=77)
    s2, ell, w = 0.18, np.array([2.99, 2.91, 1.74, 2.27, 2.01, 1.58, 1.53, 2.05]), 1e4
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_kernel(s2, ell, 0.0)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        dev.set_data(X, y)
        dev.sweep()
        P_all, B_all, _ = dev.stats()
        mu, Sig, Uv = dev.posterior()
        dev.set_data(X[:17003], y[:17003]); dev.sweep_local(); Pa, Ba, _ = dev.stats()
        dev.set_data(X[17003:], y[17003:]); dev.sweep_local(); Pb, Bb, _ = dev.stats()
        perm = np.random.default_rng(0).permutation(N)
        dev.set_data(X[perm], y[perm]); dev.sweep_local(); Pp, Bp, _ = dev.stats()
    cond_L = np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2)
    tol = min(1e-5, max(1e-9, 20 * np.finfo(float).eps * cond_L))
    assert relF(P_all, ref.stats.Psi2) < 1e-12 and relF(B_all, ref.stats.b) < 1e-12
    assert relF(mu, ref.mu_v) < tol and relF(Sig, ref.Sigma_v) < tol and relF(Uv, ref.Uv) < tol
    assert relF(Pa + Pb, P_all) < 1e-13 and relF(Ba + Bb, B_all) < 1e-13          # additivity (the all-reduce contract)
    assert relF(Pp, P_all) < 1e-13 and relF(Bp, B_all) < 1e-13                    # permutation invariance
    # one million points: no oracle at this size, only invariants
    N2, M2 = 1_000_000, 256
    rng = np.random.default_rng(5)
    X2 = rng.uniform(-1.745, 1.745, (N2, D))
    y2 = np.sin(X2.sum(axis=1))
    Xu2 = X2[:M2].copy()
    with G.SGPDevice(N2, M2, D) as dev:
        dev.set_inducing(Xu2); dev.set_data(X2, y2); dev.set_kernel(s2, ell, 1e-8)
        dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
        dev.sweep()
        P2, B2, sc = dev.stats()
        mu2, Sig2, Uv2 = dev.posterior()
    assert sc[2] == N2
    Lam2 = np.eye(M2) / 50.0 + 10.0 * P2
    assert np.abs(Lam2 @ Sig2 - np.eye(M2)).max() < 1e-6
    np.testing.assert_allclose(Sig2 @ (10.0 * B2[:, 0]), mu2, rtol=1e-6, atol=1e-9)
    assert relF(Uv2.T @ Uv2, Sig2 + np.outer(mu2, mu2)) < 1e-10
    # a 4000-point subsample of Psi2's definition (linearity check of the streaming SYRK at scale)
    idx = rng.choice(N2, 4000, replace=False)
    K = O.kernelmatrix(s2, ell, Xu2, X2[idx])
    assert np.all(np.diag(P2) >= np.sum(K * K, axis=1) - 1e-9)


def test_large_m_and_empty_data(G):
    """M well past the north-star size (18 and 32 tile rows: more Cholesky steps, inverse-factor rows and slab tiles than any
    other test) against the oracle, and the empty batch: with no data the posterior is the prior, every sum zero."""
    for N, M, D in [(3000, 1100, 3), (2000, 2048, 4)]:
        rng = np.random.default_rng(3)
        X = rng.uniform(-1.7, 1.7, (N, D))
        Xu = rng.uniform(-1.7, 1.7, (M, D))
        y = np.sin(X.sum(axis=1))
        s2, ell, w = 0.8, np.full(D, 0.6), 50.0
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(s2, ell, 1e-6)
            dev.set_prior_isotropic(50.0); dev.set_noise([[w]])
            dev.sweep()
            mu, Sig, Uv = dev.posterior()
            sc = dev.scalars()
        ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-6, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
        assert relF(mu, ref.mu_v) < 1e-7 and relF(Sig, ref.Sigma_v) < 1e-7 and relF(Uv, ref.Uv) < 1e-7
        assert math.isclose(sc.sum_I2, ref.sum_I2, rel_tol=1e-7)
        assert abs(sc.sum_I1 - ref.sum_I1) < 1e-4 * N * s2          # cancels against s_kk: cond(Kuu) * eps * s_kk
    M = 100
    with G.SGPDevice(1, M, 2) as dev:
        dev.set_inducing(np.random.default_rng(0).uniform(-1, 1, (M, 2)))
        dev.set_data(np.zeros((0, 2)), np.zeros(0))
        dev.set_kernel(1.0, np.ones(2), 1e-6); dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
        dev.sweep()
        mu, Sig, Uv = dev.posterior()
        sc = dev.scalars()
    assert np.all(mu == 0.0) and np.abs(Sig - 50.0 * np.eye(M)).max() < 1e-12
    assert np.abs(Uv - math.sqrt(50.0) * np.eye(M)).max() < 1e-12
    assert sc.sum_I1 == 0.0 and sc.sum_I2 == 0.0 and sc.energy == 0.0


# ------------------------------------------------------------------------------------------------
# The opt-in persistent factorisation launch (csrc/sgp_chain.hip.h, SGP_FLAG_PERSISTENT_CHAIN / SGP_CHAIN=persistent):
# one critical workgroup keeps the diagonal and sub-diagonal tiles in LDS across the steps, helper workgroups feed it through
# sentinel-tagged mailboxes.  Slower than the launch-per-step default on MI355X but it must stay right.
@needs_chain
@pytest.mark.parametrize("n", [1, 64, 100, 192, 300, 512, 700])
def test_persistent_chain_potrf_potri(G, n, monkeypatch):
    monkeypatch.setenv("SGP_CHAIN", "persistent")
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    A = A @ A.T / n + np.eye(n)
    for _ in range(3):                               # (hand-off races show up as run-to-run differences)
        L = G.potrf(A, variant="chain")              # (the variant library built with -DSGP_WITH_PERSISTENT_CHAIN)
        np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-12)
        Ai = G.potri(A, variant="chain")
        assert relF(Ai, np.linalg.inv(A)) < 1e-11


def test_default_library_has_no_persistent_chain(G):
    # the experiment is compiled into the variant library only: the product library refuses the flag instead of ignoring it
    from gaussianprocessnode_amd import _lib
    import ctypes as C
    lib = _lib.load()
    cfg = _lib.Config(n_max=10, m=8, d=1, d_out=1, device=0, flags=_lib.SGP_FLAG_PERSISTENT_CHAIN)
    h = C.c_void_p()
    assert lib.sgp_create(C.byref(cfg), C.byref(h)) == -1
    assert b"persistent" in lib.sgp_last_error(None)


@needs_chain
def test_persistent_chain_sweep_matches_oracle_and_is_deterministic(G):
    N, M, D, w = 1500, 512, 8, 1e4
    X, Xu, y, _ = synth(N, M, D, seed=11)
    s2, ell = 0.9, np.linspace(1.5, 3.0, D)
    with G.SGPDevice(N, M, D, persistent_chain=True) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(s2, ell, 0.0)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        dev.sweep()
        first = dev.posterior()
        KuuL = dev.kuu_chol()
        for _ in range(20):
            dev.sweep()
        again = dev.posterior()
        sc = dev.scalars()
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=0.0, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    cond_L = np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2)
    tol = min(1e-5, max(1e-9, 20 * np.finfo(float).eps * cond_L))
    assert relF(KuuL, ref.KuuL) < 1e-9
    assert relF(again[0], ref.mu_v) < tol and relF(again[1], ref.Sigma_v) < tol and relF(again[2], ref.Uv) < tol
    assert abs(sc.energy - ref.energy) <= max(1e-7, tol) * abs(ref.energy) + 1e-6
    assert sc.info_kuu == 0 and sc.info_lambda == 0


# ------------------------------------------------------------------------------------------------
# Regression tests for state the C ABI exposes freely (round-1 advisor findings)
def test_set_prior_meancov_after_a_sweep_leaves_the_posterior_alone(G):
    N, M, D = 400, 70, 2
    X, Xu, y, _ = synth(N, M, D, seed=5)
    rng = np.random.default_rng(5)
    B = rng.normal(size=(M, M))
    S0 = B @ B.T / M + np.eye(M)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.9, np.array([1.5, 2.0]), 1e-8)
        dev.set_prior_isotropic(50.0); dev.set_noise([[20.0]])
        dev.sweep()
        before = dev.posterior()
        pred_before = dev.predict(X[:50])
        dev.set_prior_meancov(rng.normal(size=M), S0)        # factors S0 on the device: must not use the results as scratch
        after = dev.posterior()
        pred_after = dev.predict(X[:50])
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    assert np.array_equal(pred_before, pred_after)


def test_predict_between_set_noise_and_theta_objective_does_not_change_the_gradient(G):
    N, M, D = 600, 96, 3
    X, Xu, y, _ = synth(N, M, D, seed=8)
    ell = np.array([1.2, 1.7, 2.2])
    res = []
    for with_predict in (False, True):
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.8, ell, 1e-8)
            dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
            dev.sweep()
            dev.set_noise([[35.0]])                          # the notebooks pass the UPDATED q(w) to the objective
            if with_predict:
                dev.predict(X[:20])
            res.append(dev.theta_objective(want_grad=True))
    (v0, g0), (v1, g1) = res
    assert v0 == v1 and np.array_equal(g0, g1)
    # and it is the gradient at w = 35 of the oracle objective (central differences)
    ref = O.vmp_sweep(Xu, X, y, None, 0.8, ell, 10.0, jitter=1e-8, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    f = lambda s2, l: O.theta_objective(Xu, X, y, s2, l, ref.mu_v, ref.Uv, 35.0, jitter=1e-8)
    h = 1e-5
    num = [(f(0.8 + h, ell) - f(0.8 - h, ell)) / (2 * h)]
    for d in range(D):
        e = np.zeros(D); e[d] = h
        num.append((f(0.8, ell + e) - f(0.8, ell - e)) / (2 * h))
    np.testing.assert_allclose(g1, num, rtol=2e-5, atol=1e-6 * abs(v1))


def test_theta_objective_gradient_length_follows_the_kernel(G):
    N, M, D = 200, 40, 3
    X, Xu, y, _ = synth(N, M, D, seed=2)
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
        dev.set_kernel(0.8, np.array([1.5]), 1e-8)           # isotropic: 1 + 1 entries
        dev.sweep()
        _, g = dev.theta_objective(want_grad=True)
        assert g.shape == (2,) and np.all(np.isfinite(g))
        with pytest.raises(ValueError):
            dev.theta_objective(want_grad=True, n_ell=3)
        dev.set_kernel(0.8, np.array([1.5, 1.6, 1.7]), 1e-8)  # ARD: 1 + D
        dev.sweep()
        _, g = dev.theta_objective(want_grad=True)
        assert g.shape == (4,) and np.all(np.isfinite(g))


def test_overlapped_sweep_with_one_reduce_per_statistics_group(G):
    """The data-sharded sweep in the OVERLAPPED order (round 4): with an all-reduce hook and the library's own streams the
    statistics are reduced group by group -- the hook is called once per tile-row group, each time with a contiguous piece of the
    exchange buffer on the stream that group runs on (group 0: [its tiles | B | scalars] on the sweep's stream, in front of the
    Lambda chain; the masked group: its tiles on the CU-masked stream, beside the chain), and the chain step that forms a masked
    group waits for an event behind that group's reduce.  On one GPU the hook adds the other shard's pieces, captured from a
    second handle.  Result: the two-shard posterior of the whole data set (GPnode/UniSGPnode.jl:62-63: the product is a sum)."""
    torch = pytest.importorskip("torch")
    from gaussianprocessnode_amd.distributed import device_tensor
    N, M, D, w = 20000, 512, 8, 1e4
    X, Xu, y, _ = synth(N, M, D, seed=31)
    s2, ell = 0.9, np.linspace(1.5, 3.0, D)
    cut = N // 2

    def make(sl):
        d = G.SGPDevice(N, M, D)
        d.set_inducing(Xu); d.set_data(X[sl], y[sl]); d.set_kernel(s2, ell, 0.0)
        d.set_prior_isotropic(50.0); d.set_noise([[w]])
        return d
    other, mine = make(slice(cut, N)), make(slice(0, cut))
    assert len(mine.overlap_plan()) >= 2             # the shard qualifies for the overlapped order, hook or not
    captured, streams = [], []

    def capture(ptr, n, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            captured.append(device_tensor(ptr, n).clone())
        streams.append(stream)
    other.set_allreduce(capture)
    plan = other.overlap_plan()                      # (re-planned when the hook is installed: every group is a collective -> one cut)
    assert len(plan) == 2, plan
    T = (M + 63) // 64
    tail = T * 64 + 8 + 1
    sizes = [g["tiles"] * 4096 + (tail if i == 0 else 0) for i, g in enumerate(plan)]
    other.sweep()
    torch.cuda.synchronize()
    assert [c.numel() for c in captured] == sizes and len(set(streams)) == len(plan)      # one call per group, each on its own stream
    calls = []

    def hook(ptr, n, stream):
        k = len(calls) % len(plan)
        calls.append(n)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            device_tensor(ptr, n).add_(captured[k])
    mine.set_allreduce(hook)
    outs = []
    for _ in range(3):
        mine.sweep()
        outs.append(mine.posterior())
    sc = mine.scalars()
    Psi2, B, scal = mine.stats()
    mine.close(); other.close()
    assert calls == sizes * 3
    for a, b in zip(outs[0], outs[2]):
        assert np.array_equal(a, b)                   # bitwise reproducible sweep to sweep
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=0.0, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    assert relF(Psi2, ref.stats.Psi2) < 1e-13 and np.array_equal(Psi2, Psi2.T) and scal[2] == N
    assert relF(B, np.reshape(ref.stats.b, B.shape)) < 1e-13
    tol = post_tol(np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2))
    mu, Sig, Uv = outs[0]
    assert relF(mu, ref.mu_v) < tol and relF(Sig, ref.Sigma_v) < tol and relF(Uv, ref.Uv) < tol
    assert abs(sc.energy - ref.energy) <= max(1e-7, tol) * abs(ref.energy) + 1e-6


def test_sweep_with_the_allreduce_hook_inside_the_library(G):
    """The C ABI's multi-GPU form (sgp_set_allreduce): ONE sgp_sweep call = local statistics -> hook -> replicated tail.  On one GPU
    the hook adds the exchange buffer of the other shard (captured beforehand from a second handle's sweep) -- what a
    sum-all-reduce over two ranks leaves in the buffer -- on the sweep's own stream.  The buffer the hook sees is the library's
    exchange buffer: lower tiles of Psi2, B, scalars (include/sgp_hip.h)."""
    torch = pytest.importorskip("torch")
    from gaussianprocessnode_amd.distributed import device_tensor
    N, M, D = 900, 80, 2
    X, Xu, y, _ = synth(N, M, D, seed=22)
    s2, ell, w = 1.0, np.array([1.3, 0.8]), 50.0
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-8, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    cut = 350

    def make(sl):
        d = G.SGPDevice(N, M, D)
        d.set_inducing(Xu); d.set_data(X[sl], y[sl]); d.set_kernel(s2, ell, 1e-8)
        d.set_prior_isotropic(50.0); d.set_noise([[w]])
        return d
    other, mine = make(slice(cut, N)), make(slice(0, cut))
    tstream = torch.cuda.Stream()
    torch.cuda.synchronize()
    T = (M + 63) // 64
    packed = T * (T + 1) // 2 * 4096 + T * 64 + 8 + 1            # lower tiles | B (Mp) | scalars (8) | Ryy (1)
    captured = []

    def capture(ptr, n, stream):
        assert n == packed and stream == tstream.cuda_stream
        with torch.cuda.stream(tstream):
            captured.append(device_tensor(ptr, n).clone())
    other.set_allreduce(capture)
    other.sweep(tstream.cuda_stream)
    torch.cuda.synchronize()
    calls = []

    def hook(ptr, n, stream):
        calls.append(n)
        if n != packed:
            return                               # (the theta gradient's data half below: left as this shard's)
        assert stream == tstream.cuda_stream
        with torch.cuda.stream(tstream):
            device_tensor(ptr, n).add_(captured[0])             # ordered on the sweep's stream, like ncclAllReduce would be
    mine.set_allreduce(hook)
    for _ in range(3):                           # the buffer is rebuilt by every sweep: no accumulation across sweeps
        mine.sweep(tstream.cuda_stream)
    torch.cuda.synchronize()
    assert calls == [packed] * 3
    mu, Sig, _ = mine.posterior(want_uv=False)
    sc = mine.scalars()
    Psi2, B, scal = mine.stats()
    assert relF(Psi2, ref.stats.Psi2) < 1e-13 and np.array_equal(Psi2, Psi2.T)      # expanded to the full symmetric matrix
    assert relF(B, np.reshape(ref.stats.b, B.shape)) < 1e-13 and scal[2] == N
    assert relF(mu, ref.mu_v) < 1e-8 and relF(Sig, ref.Sigma_v) < 1e-8
    assert math.isclose(sc.sum_I2, ref.sum_I2, rel_tol=1e-8)
    val, grad = mine.theta_objective(want_grad=True)         # the data half of the gradient goes through the hook too (33 doubles)
    assert calls[3:] == [33]
    mine.set_allreduce(None)                     # hook removed: the sweep is single-GPU again (this shard alone)
    mine.sweep(tstream.cuda_stream)
    torch.cuda.synchronize()
    assert len(calls) == 4
    ref1 = O.vmp_sweep(Xu, X[:cut], y[:cut], None, s2, ell, w, jitter=1e-8, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    assert relF(mine.posterior(want_uv=False)[0], ref1.mu_v) < 1e-8
    other.close(); mine.close()


def test_shader_clock_probe(G):
    import ctypes as C
    from gaussianprocessnode_amd import _lib
    v = C.c_double()
    _lib.check(_lib.load().sgp_measure_sclk_mhz(0, C.byref(v)), None, "sgp_measure_sclk_mhz")
    assert 500.0 < v.value < 3000.0
    # the same under FP64 matrix load, with the rate that back-to-back v_mfma_f64_16x16x4_f64 attain: what bench.py prices the
    # SYRK against next to the spec peak.  The loop cannot beat the arithmetic of its own clock (4 SIMDs x 32 flop/cycle per CU).
    out = (C.c_double * 4)()
    _lib.check(_lib.load().sgp_measure_clocks(0, out), None, "sgp_measure_clocks")
    sclk_mfma, tflops, sclk_fma, cus = out[0], out[1], out[2], out[3]
    assert 500.0 < sclk_mfma < 3000.0 and 500.0 < sclk_fma < 3000.0 and cus >= 64
    assert 10.0 < tflops <= cus * 4 * 32.0 * sclk_mfma * 1e6 / 1e12 * 1.02
    # the step kernel's diagnostics are compiled out of the product build
    tr = (C.c_int64 * 512)()
    _lib.check(_lib.load().sgp_get_step_trace(tr), None, "sgp_get_step_trace")
    assert not any(tr)


def test_rccl_adapter_with_a_single_rank_communicator():
    """sgp_use_rccl (include/sgp_hip.h): the library calls ncclAllReduce itself, inside sgp_sweep, on the sweep's stream.
    A one-rank communicator is what a one-GPU box can offer: the reduce is the identity, so results must be bitwise those
    of the handle without it.  Child process: RCCL state stays out of the test session (tests/rccl_single_rank.py)."""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "rccl_single_rank.py")], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_torch_nccl_allreduce_inside_the_sweep_with_one_rank():
    """The production multi-GPU path with the one rank a one-GPU box offers: torch.distributed backend "nccl" (RCCL),
    the all-reduce issued from the library's hook inside sgp_sweep, on the sweep's stream (tests/nccl_single_rank.py)."""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(os.path.dirname(__file__), "nccl_single_rank.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0 and "nccl single-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


# ------------------------------------------------------------------------------------------------
# The overlapped sweep (sgp_api.hip: plan_overlap / sweep_overlapped): the statistics arrive in groups of tile rows on two more
# streams (one of them CU-masked) while the Lambda chain already factors the tile columns it has.
def _sweep_once(G, X, Xu, y, s2, ell, w, jitter=0.0, repeats=1, prior=None):
    N, D = X.shape
    with G.SGPDevice(N, len(Xu), D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(s2, ell, jitter)
        if prior is None:
            dev.set_prior_isotropic(50.0)
        else:
            dev.set_prior_precision(*prior)
        dev.set_noise([[w]])
        plan = dev.overlap_plan()
        outs = []
        for _ in range(repeats):
            dev.sweep()
            outs.append(dev.posterior() + (dev.scalars(), dev.stats()))
    return plan, outs


@pytest.mark.parametrize("cols", [None, "1", "2", "3", "5", "7", "1,2", "2,4", "1,2,3,4,5,6,7"])
def test_overlapped_sweep_matches_oracle_for_every_grouping(G, cols, monkeypatch):
    # T-shaped problem (8 tile rows), every way of cutting the tile columns into groups -- also the ones the default never picks
    N, M, D, w = 10000, 512, 8, 1e4
    X, Xu, y, _ = synth(N, M, D, seed=7)
    s2, ell = 0.176, np.array([2.99, 2.91, 1.74, 2.27, 2.01, 1.58, 1.53, 2.05])
    monkeypatch.setenv("SGP_OVERLAP", "1")
    if cols:
        monkeypatch.setenv("SGP_OVERLAP_COLS", cols)
    plan, outs = _sweep_once(G, X, Xu, y, s2, ell, w, repeats=3)
    want = [int(c) for c in cols.split(",")] if cols else None
    assert len(plan) >= 2 and plan[0]["col_begin"] == 0 and plan[-1]["col_end"] == 8 and plan[0]["masked"] == 0
    if want:
        assert [g["col_end"] for g in plan[:-1]] == want
    assert sum(g["tiles"] for g in plan) == 36 and all(g["masked"] == 1 and g["cus"] == 192 for g in plan[1:])
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=0.0, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    tol = post_tol(np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2))
    mu, Sig, Uv, sc, (Psi2, B, scal) = outs[-1]
    assert relF(Psi2, ref.stats.Psi2) < 1e-13 and relF(B, np.reshape(ref.stats.b, B.shape)) < 1e-13
    assert relF(mu, ref.mu_v) < tol and relF(Sig, ref.Sigma_v) < tol and relF(Uv, ref.Uv) < tol
    assert abs(sc.energy - ref.energy) <= max(1e-7, tol) * abs(ref.energy) + 1e-6
    for a, b in zip(outs[0][:3], outs[-1][:3]):          # run-to-run bitwise identical
        assert np.array_equal(a, b)
    assert outs[0][3].energy == outs[-1][3].energy


def test_overlapped_sweep_agrees_with_the_plain_order_and_with_a_dense_prior(G, monkeypatch):
    # the same sweep with SGP_OVERLAP=0: the statistics agree to rounding (same kernels, other point chunks), the
    # posterior agrees to rounding (the tiles collect their rank-64 updates in another order); dense prior = the minibatch carry
    N, M, D, w = 6000, 320, 4, 300.0
    X, Xu, y, _ = synth(N, M, D, seed=21)
    s2, ell = 0.8, np.array([1.1, 2.0, 1.4, 0.9])
    rng = np.random.default_rng(5)
    Q = rng.normal(size=(M, M))
    L0 = Q @ Q.T / M + 0.05 * np.eye(M)
    xi0 = rng.normal(size=M)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SGP_OVERLAP", mode)
        res[mode] = _sweep_once(G, X, Xu, y, s2, ell, w, jitter=1e-8, prior=(xi0, L0))
    assert len(res["1"][0]) >= 2 and res["0"][0] == []
    a, b = res["1"][1][0], res["0"][1][0]
    assert relF(a[4][0], b[4][0]) < 1e-14 and relF(a[4][1], b[4][1]) < 1e-14       # Psi2, B: other point chunks, same sums
    ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-8, Lambda0=L0, xi0=xi0)
    tol = post_tol(np.linalg.cond(L0 + w * ref.stats.Psi2))
    for i, r in enumerate((ref.mu_v, ref.Sigma_v, ref.Uv)):
        assert relF(a[i], r) < tol and relF(b[i], r) < tol and relF(a[i], b[i]) < tol
    assert math.isclose(a[3].energy, b[3].energy, rel_tol=1e-6)     # (the energy cancels against s_kk: cond(K_uu) * eps)


def test_slab_store_switch_changes_no_bit(G, monkeypatch):
    # SGP_SYRK_WT (A/B switch of k_syrk_stream: slabs stored past the L2): same sums, same bits, in the masked groups or in all
    N, M, D, w = 6000, 320, 4, 300.0
    X, Xu, y, _ = synth(N, M, D, seed=23)
    s2, ell = 0.8, np.array([1.1, 2.0, 1.4, 0.9])
    monkeypatch.setenv("SGP_OVERLAP", "1")
    res = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("SGP_SYRK_WT", mode)
        res[mode] = _sweep_once(G, X, Xu, y, s2, ell, w, jitter=1e-8)
    base = res["0"][1][0]
    for mode in ("1", "2"):
        other = res[mode][1][0]
        assert np.array_equal(base[4][0], other[4][0]) and np.array_equal(base[4][1], other[4][1])      # Psi2, B
        for i in range(3):
            assert np.array_equal(base[i], other[i])                                                    # mu_v, Sigma_v, Uv


def test_overlapped_and_plain_sweeps_interleave_on_one_handle(G, monkeypatch):
    # sweep() (overlapped) and sweep_local() + sweep_finish() (plain order, the two-phase entry points of the multi-GPU path)
    # on the same handle, back to back without waiting in between: the done word / statistics words keep them apart
    N, M, D, w = 5000, 256, 8, 1e3
    X, Xu, y, _ = synth(N, M, D, seed=3)
    monkeypatch.setenv("SGP_OVERLAP", "1")
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(1.0, np.full(D, 2.0), 0.0)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        assert len(dev.overlap_plan()) >= 2
        for _ in range(4):
            dev.sweep()
            dev.sweep_local()
            dev.sweep_finish()
        plain = dev.posterior()
        dev.sweep()
        over = dev.posterior()
        sc = dev.scalars()
    ref = O.vmp_sweep(Xu, X, y, None, 1.0, np.full(D, 2.0), w, jitter=0.0, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
    tol = post_tol(np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2))
    for a, b, r in zip(plain, over, (ref.mu_v, ref.Sigma_v, ref.Uv)):
        assert relF(a, r) < tol and relF(b, r) < tol
    assert abs(sc.energy - ref.energy) <= max(1e-7, tol) * abs(ref.energy) + 1e-6


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_a_bounded_wait_that_gives_up_is_reported_not_swallowed(G, overlap, monkeypatch):
    # SGP_SPIN_LIMIT=1 (test hook): every device-word wait gives up at its first unsuccessful poll.  The second of two sweeps
    # queued back to back starts its K_uu chain / statistics before the first has finished -- exactly the hand-off the words
    # protect -- so the getters must refuse the results (SGP_ERR_HIP naming the word); after that the handle is usable again.
    from gaussianprocessnode_amd._lib import SGPError
    N, M, D, w = 10000, 512, 8, 1e4
    X, Xu, y, _ = synth(N, M, D, seed=9)
    monkeypatch.setenv("SGP_OVERLAP", overlap)
    monkeypatch.setenv("SGP_SPIN_LIMIT", "1")
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, y)
        dev.set_kernel(1.0, np.full(D, 2.0), 0.0)
        dev.set_prior_isotropic(50.0)
        dev.set_noise([[w]])
        for _ in range(3):
            dev.sweep()
        with pytest.raises(SGPError, match="bounded device-word wait gave up"):
            dev.scalars()
        # sticky: EVERY getter refuses what that sweep left behind, not only the first one asked
        for getter in (dev.posterior, dev.stats, dev.kuu_chol, dev.theta_objective):
            with pytest.raises(SGPError, match="bounded device-word wait gave up"):
                getter()
        # the next sweep clears the word: the handle is usable again (with a one-poll limit a sweep's own hand-offs
        # -- statistics groups -> Lambda chain, K_uu chain -> Sigma launch -- may give up too, which is reported the same way)
        dev.sweep()
        try:
            assert np.isfinite(dev.scalars().energy)
        except SGPError as e:
            assert "bounded device-word wait gave up" in str(e)


# ------------------------------------------------------------------------------------------------
# Process exit with a live handle (SIGSEGV inside __cxa_finalize under rocprofv3 when a script left a handle with a
# registered ctypes all-reduce hook behind).  Teardown must not depend on the caller's tidiness: the Python mirror closes live
# handles in an atexit handler, the library destroys what is still registered in an exit handler of its own.
@pytest.mark.parametrize("python_atexit", [True, False], ids=["python-atexit", "library-exit-handler"])
def test_process_exit_with_a_live_hooked_handle_is_clean(python_atexit):
    import subprocess
    import sys
    code = f"""
import atexit, sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import numpy as np, torch
from gaussianprocessnode_amd import device
from gaussianprocessnode_amd.distributed import HipEngine
if not {python_atexit}:
    atexit.unregister(device._close_all_at_exit)      # leave the handle to the library's own exit handler
rng = np.random.default_rng(0)
N, M, D = 2000, 128, 3
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = np.sin(X.sum(1))
eng = HipEngine(N, M, D, 1, device=0)
eng.install_allreduce(lambda t: t.mul_(1.0))        # a hook: the closure and the engine reference each other
dev = eng.dev
dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.9, np.full(D, 1.5), 1e-8)
dev.set_prior_isotropic(50.0); dev.set_noise([[100.0]])
for _ in range(5):
    eng.sweep()
print("energy", dev.scalars().energy, flush=True)
eng.sweep()                                          # ... and one more sweep still in flight at exit
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "energy" in r.stdout
