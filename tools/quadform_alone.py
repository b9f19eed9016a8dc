#!/usr/bin/env python3
"""k_quadform_fused alone at T (20 launches), for counter passes:  rocprofv3 --kernel-trace --pmc ... -- python3 tools/quadform_alone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGP_OVERLAP", "0")     # (counter passes serialise the kernels: the plain order has no cross-stream words to wait for)
import bench
from gaussianprocessnode_amd import SGPDevice, _lib
N, M, D = 10000, 512, 8
X, Xu, y, _, _ = bench.synthetic(N, M, D)
with SGPDevice(N, M, D, keep_kuf=True) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    dev.sweep(); dev.scalars()
    print("quadform alone:", dev.time_kernel(_lib.SGP_TIME_QUADFORM, 20), "us")
