#!/usr/bin/env python3
"""Sweep rate and streaming-SYRK throughput against the number of points at M = 512, D = 8 (one MI355X): where the
data-sized kernels take over from the latency-bound M^3 tail (SURVEY.md §8d scaling sweep)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd import _lib

M, D = 512, 8
ELL = np.array([2.99, 2.91, 1.74, 2.27, 2.01, 1.58, 1.53, 2.05])
print(f"{'N':>9s} {'sweeps/s':>10s} {'ms/sweep':>9s} {'Mpoints/s':>10s} {'gram us':>9s} {'syrk us':>9s} {'syrk TF/s':>10s} {'tail us':>8s}")
for N in (10_000, 100_000, 1_000_000, 4_000_000):
    rng = np.random.default_rng(1)
    X = rng.uniform(-1.745, 1.745, (N, D)); y = np.sin(X.sum(1)); Xu = X[:M].copy()
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.176, ELL, 1e-8)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        reps = 100 if N <= 100_000 else 20
        for _ in range(3): dev.sweep()
        dev.scalars(); dev.phase_totals(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps): dev.sweep()
        dev.scalars()
        dt = (time.perf_counter() - t0) / reps
        ph, _ = dev.phase_totals()
        syrk = ph[_lib.SGP_T_SYRK]
        print(f"{N:9d} {1/dt:10.1f} {1e3*dt:9.3f} {N/dt/1e6:10.1f} {ph[_lib.SGP_T_GRAM]:9.1f} {syrk:9.1f} "
              f"{N*M*(M+1)/(syrk*1e-6)/1e12:10.2f} {ph[_lib.SGP_T_FINISH1]+ph[_lib.SGP_T_FINISH2]:8.1f}", flush=True)
