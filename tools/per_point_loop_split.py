#!/usr/bin/env python3
"""Per-point loop (sgp_sweep; sgp_w_stats) at T and C4: where an iteration's wall time goes on the host (the enqueue of the sweep,
the blocking call), eager (overlapped order) against a captured graph (plain order), and the sweep alone after a wait."""
import os, sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocessnode_amd import SGPDevice
for name, N, M, D in (("T", 10000, 512, 8), ("C4", 4000, 128, 2)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
    for graph in (False, True):
        with SGPDevice(N, M, D, keep_kuf=True, use_graph=graph) as dev:
            dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
            dev.set_prior_isotropic(50.0); dev.set_noise(np.eye(1) * 10.0)
            for _ in range(30):
                dev.sweep(); dev.w_stats()
            for rep in range(3):
                a = b = 0.0
                t0 = time.perf_counter()
                for _ in range(200):
                    t1 = time.perf_counter(); dev.sweep(); t2 = time.perf_counter(); dev.w_stats(); t3 = time.perf_counter()
                    a += t2 - t1; b += t3 - t2
                tot = time.perf_counter() - t0
                # the sweep alone, waited for every time (what the loop's sweep costs when the device starts idle)
                t0 = time.perf_counter()
                for _ in range(200):
                    dev.sweep(); dev.wait()
                alone = time.perf_counter() - t0
                print(f"{name} {'graph' if graph else 'eager'}: {200 / tot:7.1f} it/s = {1e6 * tot / 200:6.1f} us (sweep call {1e6 * a / 200:6.1f}, w_stats call "
                      f"{1e6 * b / 200:6.1f}); sweep + wait alone {1e6 * alone / 200:6.1f} us", flush=True)
