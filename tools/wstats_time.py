#!/usr/bin/env python3
"""Where the time of the per-point path goes: wall time of sgp_sweep + sgp_get_scalars, of sgp_w_stats behind a
finished sweep, and of the two in a loop, at T and at the banana shape.    python tools/wstats_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
_spec = importlib.util.spec_from_file_location("_sgp_hostbind", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocessnode_amd", "hostbind.py"))
_hb = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_hb)
print("host binding:", _hb.bind_to_gpu_node(0))       # (before the HIP runtime starts: see hostbind.py)
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd import _lib

for name, N, M, D in (("T", 10000, 512, 8), ("C4", 4000, 128, 2)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
    with G.SGPDevice(N, M, D, keep_kuf=True) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
        dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
        for _ in range(20):
            dev.sweep(); dev.w_stats()
        # every figure: the median of 5 blocks of 40 (the host thread of a shared box loses a 10 ms tick now and then, A/B log [35])
        def timed(body, blocks=5, reps=40):
            ts = []
            for _ in range(blocks):
                t0 = time.perf_counter()
                for _ in range(reps):
                    body()
                ts.append((time.perf_counter() - t0) / reps)
            return float(np.median(ts))
        t_sweep = timed(lambda: (dev.sweep(), dev.scalars()))
        dev.sweep(); dev.scalars()
        t_w = timed(dev.w_stats)
        t_both = timed(lambda: (dev.sweep(), dev.w_stats()))
        q = dev.time_kernel(_lib.SGP_TIME_QUADFORM, 20)
        print(f"{name}: sweep + get_scalars (one at a time) {1e6 * t_sweep:.1f} us | w_stats alone {1e6 * t_w:.1f} us (the quadratic-form kernel {q:.1f} us) | "
              f"sweep + w_stats {1e6 * t_both:.1f} us = {1 / t_both:.0f} it/s")
