#!/usr/bin/env python3
"""Back-to-back sweep rate with hipGraph replay vs eager launches, for a few problem sizes (one MI355X)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gaussianprocessnode_amd as G

for N, M, D in [(50, 20, 1), (200, 64, 2), (1000, 128, 2), (2000, 256, 8), (500, 600, 8), (10000, 512, 8)]:
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
    out = []
    for use_graph in (True, False):
        with G.SGPDevice(N, M, D, use_graph=use_graph) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.ones(D), 1e-6)
            dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
            for _ in range(20): dev.sweep()
            dev.scalars()
            t0 = time.perf_counter()
            for _ in range(300): dev.sweep()
            dev.scalars()
            out.append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"N={N:6d} M={M:4d} D={D}: graph {out[0]:8.1f} us/sweep   eager {out[1]:8.1f} us/sweep")
