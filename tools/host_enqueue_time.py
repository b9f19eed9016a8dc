#!/usr/bin/env python3
"""Host time of sgp_sweep at T: how long does the host need to enqueue one sweep, and how does a block of 20 sweeps unfold?"""
import os, sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussianprocessnode_amd import SGPDevice
N, M, D = 10000, 512, 8
X, Xu, y, _, _ = bench.synthetic(N, M, D)
with SGPDevice(N, M, D) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    for _ in range(30): dev.sweep()
    dev.wait()
    for rep in range(5):
        dev.wait()
        ts = [time.perf_counter()]
        for _ in range(20):
            dev.sweep(); ts.append(time.perf_counter())
        t_enq = ts[-1] - ts[0]
        dev.wait(); t_end = time.perf_counter() - ts[0]
        per = np.diff(ts) * 1e6
        print(f"block of 20: host enqueue {1e6 * t_enq:7.1f} us ({1e6 * t_enq / 20:5.1f} per sweep; first {per[0]:5.1f}, median {np.median(per):5.1f}, max {per.max():5.1f}), "
              f"all done after {1e6 * t_end:7.1f} us = {1e6 * t_end / 20:5.1f} per sweep", flush=True)
