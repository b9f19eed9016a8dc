#!/usr/bin/env python3
"""The overlap planner's choice (tile-column groups of the overlapped sweep) for the bench workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussianprocessnode_amd import SGPDevice
for name, (N, M, D) in bench.WORKLOADS.items():
    if N > 200000: continue
    X, Xu, y, _, _ = bench.synthetic(N, M, D)
    with SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL[:D], 0.0)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        plan = dev.overlap_plan()
        print(name, N, M, [(g['col_begin'], g['col_end'], g['chunks'], g['points_per_chunk']) for g in plan] if plan else "plain order", flush=True)
