#!/usr/bin/env python3
"""What do the host copies of sgp_w_stats cost?  sweep + w_stats per iteration with the results copied into fresh arrays (the mirror's
w_stats()), into preallocated arrays, and not copied at all (NULL outputs: the values stay in the pinned block)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussianprocessnode_amd import SGPDevice
from gaussianprocessnode_amd._lib import ptr
N, M, D = 10000, 512, 8
X, Xu, y, _, _ = bench.synthetic(N, M, D)
with SGPDevice(N, M, D, keep_kuf=True) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    I1, I2 = np.empty(N), np.empty(N)
    def fresh(): dev.sweep(); dev.w_stats()
    def prealloc(): dev.sweep(); dev._check(dev._lib.sgp_w_stats(dev._h, ptr(I1), ptr(I2), None), "sgp_w_stats")
    def nocopy(): dev.sweep(); dev._check(dev._lib.sgp_w_stats(dev._h, None, None, None), "sgp_w_stats")
    for name, f in (("fresh arrays", fresh), ("preallocated", prealloc), ("no host copy", nocopy), ("fresh arrays", fresh), ("no host copy", nocopy)):
        for _ in range(30): f()
        t0 = time.perf_counter()
        for _ in range(300): f()
        dt = (time.perf_counter() - t0) / 300
        print(f"{name:14s} {1e6 * dt:7.1f} us per iteration = {1 / dt:6.0f} it/s", flush=True)
    sys.exit(0) if os.environ.get("WSTATS_SHORT") else None
    # the slow mode of a preallocated destination: which destination addresses trigger it?
    import ctypes
    for off in (0, 1, 2, 6, 8, 64, 510, 512, 1024):
        base1, base2 = np.zeros(N + 2048), np.zeros(N + 2048)
        a1, a2 = base1[off:off + N], base2[off:off + N]
        def pre2(): dev.sweep(); dev._check(dev._lib.sgp_w_stats(dev._h, ptr(a1), ptr(a2), None), "sgp_w_stats")
        for _ in range(30): pre2()
        t0 = time.perf_counter()
        for _ in range(200): pre2()
        dt = (time.perf_counter() - t0) / 200
        print(f"preallocated, offset {off:5d} doubles (address mod 4096 = {a1.ctypes.data % 4096:4d} / {a2.ctypes.data % 4096:4d}): {1e6 * dt:7.1f} us per iteration", flush=True)
