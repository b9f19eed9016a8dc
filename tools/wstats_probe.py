#!/usr/bin/env python3
"""Why does the sweep + sgp_w_stats loop take 460 us per iteration inside bench.py and 630 - 780 us in tools/wstats_time.py?  The same
loop at T under: nothing else (a), torch's CUDA context initialised first (b), a second idle handle alive (c), both (d).
    python tools/wstats_probe.py a|b|c|d"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode in "bd":
    import torch
    torch.cuda.set_device(0)
    keep = torch.zeros(16, device="cuda")
    torch.cuda.synchronize()
import gaussianprocessnode_amd as G
N, M, D = 10000, 512, 8
rng = np.random.default_rng(0)
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = rng.uniform(-1.7, 1.7, (M, D)); y = np.sin(X.sum(1))
other = G.SGPDevice(64, 16, 2) if mode in "cd" else None
with G.SGPDevice(N, M, D, keep_kuf=True) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
    dev.set_prior_isotropic(50.0); dev.set_noise([[10.0]])
    for _ in range(20):
        dev.sweep(); dev.w_stats()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            dev.sweep(); dev.w_stats()
        out.append((time.perf_counter() - t0) / 200)
    t0 = time.perf_counter()
    for _ in range(200):
        dev.sweep(); dev.scalars()
    t_s = (time.perf_counter() - t0) / 200
print(f"mode {mode}: sweep + w_stats " + " ".join(f"{1e6 * t:.1f}" for t in out) + f" us per iteration; sweep + get_scalars {1e6 * t_s:.1f} us")
if other is not None:
    other.close()
