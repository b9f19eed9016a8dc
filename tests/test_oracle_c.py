"""The plain-C per-point restatement (reference control flow) agrees with the batched NumPy oracle."""
import math

import numpy as np
import pytest

from oracle import c_oracle
from oracle import sgp_oracle as O


@pytest.mark.parametrize("N,M,D,cls", [(40, 9, 1, False), (120, 33, 3, True), (3, 6, 2, False)])
def test_c_perpoint_equals_numpy_batched(N, M, D, cls):
    rng = np.random.default_rng(N + M)
    X, Xu, y = rng.uniform(-2, 2, (N, D)), rng.uniform(-2, 2, (M, D)), rng.normal(size=N)
    vy = rng.uniform(0.1, 0.4, N) if cls else None
    s2, ell, w, jit = 0.8, rng.uniform(1.0, 2.0, D), 30.0, 1e-8
    A = rng.normal(size=(M, M))
    Sigma0, mu0 = A @ A.T / M + np.eye(M), 0.1 * rng.normal(size=M)
    E_logw = math.log(w) - 0.02
    c = c_oracle.vmp_sweep_perpoint(Xu, X, y, vy, s2, ell, jit, w, E_logw, mu0, Sigma0, want_points=True)
    r = O.vmp_sweep(Xu, X, y, vy, s2, ell, w, E_logw=E_logw, jitter=jit, mu0=mu0, Sigma0=Sigma0)
    np.testing.assert_allclose(c["mu_v"], r.mu_v, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(c["Sigma_v"], r.Sigma_v, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(c["Uv"], r.Uv, rtol=1e-8, atol=1e-11)
    I1, I2 = O.w_stats_perpoint(Xu, X, y, vy, s2, ell, r.KuuL, r.mu_v, r.Uv)
    np.testing.assert_allclose(c["I1"], I1, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(c["I2"], I2, rtol=1e-9, atol=1e-11)
    assert math.isclose(c["sum_I2"], r.sum_I2, rel_tol=1e-9)
    assert abs(c["sum_I1"] - r.sum_I1) <= 1e-7 * r.stats.s_kk
    assert math.isclose(c["energy"], r.energy, rel_tol=1e-7, abs_tol=1e-6 * w)
