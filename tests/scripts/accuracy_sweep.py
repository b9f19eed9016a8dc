#!/usr/bin/env python3
"""Accuracy sweep over many random draws of three shapes (toy: 1-D inputs, cond(K_uu) ~ 1e9): worst error of the K_uu factor
and of the posterior, in units of cond * eps, and the number of draws that would fail tests/test_gpu_parity.py's tolerances.
Usage: accuracy_sweep.py <seeds> [toy]   (run it after any change to the factorisation kernels; a variant of the pivot loop
that let the two triangles of the diagonal block drift apart passed the fixed-seed tests and failed 8 of 120 draws here)"""
import sys, math, zlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import gaussianprocessnode_amd as G
from oracle import sgp_oracle as O
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity import synth, relF
shapes_all = [("toy", 50, 20, 1, 100.0, 1e-8, False), ("ragged", 333, 37, 3, 10.0, 1e-8, False), ("mid", 700, 130, 2, 30.0, 1e-8, False)]
shapes = shapes_all[:1] if len(sys.argv) > 2 else shapes_all
eps = np.finfo(float).eps
for name, N, M, D, w, jit, cls in shapes:
    worst = {"kuu": 0, "mu": 0, "sig": 0, "uv": 0}; nfail = 0; nbad_info = 0
    for seed in range(int(sys.argv[1])):
        X, Xu, y, vy = synth(N, M, D, seed=seed, classification=cls)
        s2, ell = 0.9, np.linspace(1.5, 3.0, D)
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y, vy); dev.set_kernel(s2, ell, jit)
            dev.set_prior_isotropic(50.0); dev.set_noise([[w]], math.log(w) - 0.01)
            try:
                dev.sweep()
            except Exception as e:
                nbad_info += 1; continue
            KuuL = dev.kuu_chol(); mu, Sig, Uv = dev.posterior()
        ref = O.vmp_sweep(Xu, X, y, vy, s2, ell, w, E_logw=math.log(w) - 0.01, jitter=jit, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
        Kuu = O.kernelmatrix(s2, ell, Xu) + jit * np.eye(M)
        cK = np.linalg.cond(Kuu); cL = np.linalg.cond(np.eye(M) / 50.0 + w * ref.stats.Psi2)
        r = {"kuu": relF(KuuL, ref.KuuL) / (cK * eps), "mu": relF(mu, ref.mu_v) / (cL * eps), "sig": relF(Sig, ref.Sigma_v) / (cL * eps), "uv": relF(Uv, ref.Uv) / (cL * eps)}
        for k in worst: worst[k] = max(worst[k], r[k])
        tol_post = min(1e-5, max(1e-9, 20 * eps * cL))
        if relF(KuuL, ref.KuuL) >= 1e-9 or relF(mu, ref.mu_v) >= tol_post or relF(Sig, ref.Sigma_v) >= tol_post or relF(Uv, ref.Uv) >= tol_post: nfail += 1
    print(name, "worst error / (cond*eps):", {k: round(v, 3) for k, v in worst.items()}, "test-tolerance failures:", nfail, "info!=0:", nbad_info, flush=True)
