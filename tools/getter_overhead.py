#!/usr/bin/env python3
"""Fixed host cost of the blocking calls on an IDLE device (nothing queued): sgp_wait (three stream queries + hipDeviceSynchronize),
sgp_get_scalars (the same + the pinned mirror), sgp_w_stats' parts that are not kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _bind  # noqa: E401,E702
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocessnode_amd import SGPDevice
N, M, D = 10000, 512, 8
rng = np.random.default_rng(0)
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
with SGPDevice(N, M, D, keep_kuf=True) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
    dev.set_prior_isotropic(50.0); dev.set_noise(np.eye(1) * 10.0)
    dev.sweep(); dev.scalars()
    for name, f in (("sgp_wait", dev.wait), ("sgp_get_scalars", dev.scalars)):
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(5000): f()
            print(f"{name:16s} on an idle device: {1e6 * (time.perf_counter() - t0) / 5000:6.2f} us per call", flush=True)
