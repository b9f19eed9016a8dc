#!/usr/bin/env python3
"""Durations of the streaming SYRK launches alone (HIP events around eager launches): the single launch over all tiles at N = 10^6 and
at T, and T's two group launches.  For A/B runs of kernel variants (copy the library into place first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from gaussianprocessnode_amd import SGPDevice, _lib
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for N in (10000, 1000000):
    M, D = 512, 8
    X, Xu, y, _, _ = bench.synthetic(N, M, D)
    with SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        try:
            dev.sweep(); dev.scalars()
        except Exception as e:            # (timing variants with wrong results may fail the factorisation)
            print(' sweep:', str(e)[:80])
        one = min(dev.time_kernel(_lib.SGP_T_SYRK, 10) for _ in range(3))
        fl = N * M * (M + 1.0)
        line = f"{tag} N={N}: single launch {one:.1f} us = {fl / one * 1e-6:.1f} TFLOP/s"
        if N == 10000:
            g = [min(dev.time_kernel(_lib.SGP_TIME_GROUP0 + k, 10) for _ in range(3)) for k in range(2)]
            line += f"; groups {g[0]:.1f} / {g[1]:.1f} us"
        gram = min(dev.time_kernel(_lib.SGP_T_GRAM, 10) for _ in range(3))
        line += f"; Gram {gram:.1f} us = {8.0 * N * 512 / gram * 1e-6:.2f} TB/s of stores"
        print(line, flush=True)
