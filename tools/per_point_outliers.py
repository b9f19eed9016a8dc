#!/usr/bin/env python3
"""Per-iteration wall times of the per-point loop (sgp_sweep; sgp_w_stats) and of (sgp_sweep; sgp_wait) at T: how often an iteration
takes much longer than the median, how long, and in which call."""
import gc, os, sys, time
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _bind  # noqa: E401,E702  (NUMA node of the GPU first)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocessnode_amd import SGPDevice
N, M, D = 10000, 512, 8
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.disable(); print("cyclic garbage collector disabled")
rng = np.random.default_rng(0)
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
with SGPDevice(N, M, D, keep_kuf=True) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
    dev.set_prior_isotropic(50.0); dev.set_noise(np.eye(1) * 10.0)
    for _ in range(30):
        dev.sweep(); dev.w_stats()
    for what in ("w_stats", "w_stats", "wait"):
        n = 4000
        ts = np.empty((n, 2))
        second = dev.w_stats if what == "w_stats" else dev.wait
        t_all = time.perf_counter()
        for i in range(n):
            t1 = time.perf_counter(); dev.sweep(); t2 = time.perf_counter(); second(); t3 = time.perf_counter()
            ts[i] = (t2 - t1, t3 - t2)
        t_all = time.perf_counter() - t_all
        it = ts.sum(1) * 1e6
        med = np.median(it)
        slow = np.nonzero(it > 2 * med)[0]
        print(f"sweep; {what}: {n} iterations, mean {1e6 * t_all / n:7.1f} us, median {med:6.1f}, p99 {np.percentile(it, 99):7.1f}, max {it.max():9.1f}; "
              f"{len(slow)} iterations over twice the median, {it[slow].sum() / 1e3:7.1f} ms in them", flush=True)
        for i in slow[np.argsort(-it[slow])][:8]:
            print(f"      iteration {i:5d}: sweep call {1e6 * ts[i, 0]:9.1f} us, {what} call {1e6 * ts[i, 1]:9.1f} us")
