#!/usr/bin/env python3
"""Random shapes through k_syrk_direct (points x lower tiles >= 10 000): statistics against the oracle, with and without per-point
weights, point counts that are no multiple of 4, tile counts from 1 to 45; overlapped and plain order.   python tests/scripts/direct_syrk_sweep.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gaussianprocessnode_amd as G
from oracle import sgp_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
worst = 0.0
for c in range(cases):
    M = int(rng.choice([40, 64, 65, 100, 128, 130, 200, 256, 300, 384, 512, 600]))
    T = (M + 63) // 64
    tiles = T * (T + 1) // 2
    nmin = 10000 // tiles + 1
    N = int(rng.integers(nmin, max(nmin + 50, 4 * nmin)))
    D = int(rng.integers(1, 9))
    weighted = bool(rng.integers(0, 2))
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
    om = rng.uniform(0.2, 1.5, N) if weighted else None
    s2, ell = 0.9, rng.uniform(1.0, 2.5, D)
    st = O.suff_stats(Xu, X, y, None, s2, ell, omega=om)
    for order in ("auto", "plain"):
        if order == "plain": os.environ["SGP_OVERLAP"] = "0"
        else: os.environ.pop("SGP_OVERLAP", None)
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y, weights=om); dev.set_kernel(s2, ell, 1e-8)
            dev.set_prior_isotropic(50.0); dev.set_noise([[20.0]])
            plan = dev.overlap_plan()
            dev.sweep()
            Psi2, B, _ = dev.stats()
            sc = dev.scalars()
        e1 = np.linalg.norm(Psi2 - st.Psi2) / np.linalg.norm(st.Psi2)
        e2 = np.linalg.norm(B - st.b) / np.linalg.norm(st.b)
        worst = max(worst, e1, e2)
        flag = "" if (e1 < 1e-13 and e2 < 1e-13 and sc.info_kuu == 0 and sc.info_lambda == 0) else "   <-- FAIL"
        print(f"N={N:6d} M={M:4d} D={D} weights={int(weighted)} {order:5s} groups={len(plan) if plan else 0}: Psi2 {e1:.1e} B {e2:.1e}{flag}", flush=True)
    os.environ.pop("SGP_OVERLAP", None)
print("worst relative Frobenius error", worst)
