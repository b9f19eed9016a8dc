#!/usr/bin/env python3
"""Sweep rates of the BASELINE.json configurations on one MI355X (back-to-back sweeps, results resident)."""
import sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gaussianprocessnode_amd as G

CONFIGS = [("C1 toy 1-D", 50, 20, 1, 1), ("C2 kin40k M=256", 10000, 256, 8, 1), ("T  kin40k M=512", 10000, 512, 8, 1),
           ("C3 kin40k N=40k", 40000, 512, 8, 1), ("C4 banana", 4000, 128, 2, 1), ("C5 pendulum MultiSGP", 1500, 48, 2, 2)]
for name, N, M, D, Do in CONFIGS:
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D))
    Y = np.sin(X.sum(1))[:, None] * np.ones((1, Do))
    with G.SGPDevice(N, M, D, d_out=Do) as dev:
        dev.set_inducing(Xu)
        dev.set_data(X, Y if Do > 1 else Y[:, 0], None, np.full(N, 0.2) if Do > 1 else None, n_nodes=(N // 5 if Do > 1 else None))
        dev.set_kernel(1.0, np.full(D, 1.5), 1e-6); dev.set_prior_isotropic(50.0)
        dev.set_noise(np.eye(Do) * 10.0)
        for _ in range(20): dev.sweep()
        dev.scalars()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps): dev.sweep()
        dev.scalars()
        dt = (time.perf_counter() - t0) / reps
    print(f"{name:24s} N={N:6d} M={M:4d} D={D} d_out={Do}: {1/dt:9.1f} sweeps/s  {1e6*dt:8.1f} us/sweep", flush=True)
