#!/usr/bin/env python3
"""k_gram_uf alone at N = 10^6, M = 512 (20 launches), for counter passes:  rocprofv3 --kernel-trace --pmc ... -- python3 tools/gram_alone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussianprocessnode_amd import SGPDevice, _lib
N, M, D = int(os.environ.get("GRAM_N", 1000000)), 512, 8
X, Xu, y, _, _ = bench.synthetic(N, M, D)
with SGPDevice(N, M, D) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    dev.sweep(); dev.scalars()
    print("Gram alone:", dev.time_kernel(_lib.SGP_T_GRAM, 20), "us")
