import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussianprocessnode_amd as G
N, M, D = 500, 512, 8
rng = np.random.default_rng(0)
X = rng.normal(size=(N, D)); y = rng.normal(size=N); Xu = rng.normal(size=(M, D))
with G.SGPDevice(N, M, D) as dev:
    dev.set_inducing(Xu); dev.set_kernel(1.0, np.full(D, 2.0), 0.0); dev.set_prior_isotropic(50.0); dev.set_noise([[100.0]])
    for _ in range(20): dev.set_data(X, y)
    t0 = time.perf_counter()
    for _ in range(500): dev.set_data(X, y)
    t1 = time.perf_counter()
    print(f"set_data(n={N}): {(t1 - t0) / 500 * 1e6:.1f} us per call")
    yv = np.abs(rng.normal(size=N)); w = np.abs(rng.normal(size=N))
    t0 = time.perf_counter()
    for _ in range(500): dev.set_data(X, y, yv, w)
    t1 = time.perf_counter()
    print(f"set_data(n={N}, y_var, weights): {(t1 - t0) / 500 * 1e6:.1f} us per call")
    dev.sweep(); print(dev.scalars().energy)
