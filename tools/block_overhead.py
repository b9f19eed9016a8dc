#!/usr/bin/env python3
"""Wall time of a block of k back-to-back sweeps (wait; k x sgp_sweep; wait) against k, at T: where the fixed cost of a block accrues."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _bind  # noqa: E401,E702
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussianprocessnode_amd import SGPDevice
N, M, D = 10000, 512, 8
X, Xu, y, _, _ = bench.synthetic(N, M, D)
with SGPDevice(N, M, D) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    for _ in range(50): dev.sweep()
    dev.wait()
    prev = 0.0
    for k in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 30, 40, 80, 160):
        ts = []
        for rep in range(25):
            dev.wait()
            t0 = time.perf_counter()
            for _ in range(k): dev.sweep()
            dev.wait()
            ts.append(time.perf_counter() - t0)
        med = 1e6 * float(np.median(ts))
        print(f"k = {k:3d}: {med:8.1f} us = {med / k:6.1f} per sweep; minus k x 215.0: {med - 215.0 * k:6.1f}; since the last line {(med - prev):7.1f}", flush=True)
        prev = med
