#!/usr/bin/env python3
"""Sweep rate against the "the SYRK fills the chip" threshold (SGP_GATE_MIN: points x lower tiles from which the K_uu chain is gated
behind the SYRK, the SYRK is k_syrk_direct and the overlapped order is considered) for problem shapes around the threshold."""
import os, sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gaussianprocessnode_amd as G

SHAPES = [(4000, 128, 2), (10000, 128, 8), (1000, 512, 8), (1500, 512, 8), (4000, 256, 8), (2000, 512, 8), (3000, 512, 8), (600, 1024, 8),
          (10000, 256, 8), (5000, 512, 8)]
GATES = [200000, 100000, 50000, 25000, 10000]
print(f"{'N':>6s} {'M':>5s} {'N x tiles':>10s} " + " ".join(f"{'gate ' + str(g):>12s}" for g in GATES))
for N, M, D in SHAPES:
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
    T = (M + 63) // 64
    row = []
    for g in GATES:
        os.environ["SGP_GATE_MIN"] = str(g)
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
            dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
            for _ in range(20): dev.sweep()
            dev.scalars()
            best = 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(300): dev.sweep()
                dev.scalars()
                best = max(best, 300 / (time.perf_counter() - t0))
            plan = dev.overlap_plan()
        row.append(f"{best:8.0f}{'*' if plan else ' '}{len(plan) if plan else 0:1d}  ")
    print(f"{N:6d} {M:5d} {N * T * (T + 1) // 2:10d} " + " ".join(row), flush=True)
print("(* = overlapped order, with the number of statistics groups)")
