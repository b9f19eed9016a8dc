#!/usr/bin/env python3
"""Sweep rate with the planner's choice against the plain order (SGP_OVERLAP=0) for a list of shapes."""
import os, sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gaussianprocessnode_amd as G
SHAPES = [(10000, 256, 8), (3000, 512, 8), (2000, 512, 8), (1500, 512, 8), (1000, 512, 8), (4000, 256, 8), (600, 1024, 8), (5000, 512, 8), (10000, 192, 8),
          (2000, 1024, 8), (20000, 256, 8), (500, 600, 8)]
if len(sys.argv) > 1: SHAPES = [tuple(int(v) for v in a.split(",")) + (8,) for a in sys.argv[1:]]
print(f"{'N':>6s} {'M':>5s}  planner                                   sweeps/s   plain order")
for N, M, D in SHAPES:
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
    out = []
    for ov in (None, "0"):
        if ov is None: os.environ.pop("SGP_OVERLAP", None)
        else: os.environ["SGP_OVERLAP"] = ov
        with G.SGPDevice(N, M, D) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
            dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
            for _ in range(20): dev.sweep()
            dev.scalars()
            best, reps = 0.0, (300 if N <= 50000 else 40)
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps): dev.sweep()
                dev.scalars()
                best = max(best, reps / (time.perf_counter() - t0))
            plan = dev.overlap_plan()
        out.append((best, [(g['col_begin'], g['col_end']) for g in plan] if plan else "plain"))
    os.environ.pop("SGP_OVERLAP", None)
    print(f"{N:6d} {M:5d}  {str(out[0][1]):40s} {out[0][0]:9.0f} {out[1][0]:9.0f}   {100 * (out[0][0] / out[1][0] - 1):+.1f} %", flush=True)
