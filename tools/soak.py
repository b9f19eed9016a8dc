#!/usr/bin/env python3
"""Soak: thousands of back-to-back sweeps on the real kin40k data, every result compared bitwise with the first
(an inter-kernel hazard shows up as a run-to-run difference or a spurious PosDef failure)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd.meta import softplus

n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
d = np.load(os.path.join(ROOT, "tests", "golden", "kin40k_data.npz")); f = np.load(os.path.join(ROOT, "tests", "golden", "kin40k_fixture.npz"))
p = softplus(f["theta_opt"])
bad = 0
for M, use_graph in ((512, False), (600, False), (512, True)):
    with G.SGPDevice(10000, M, 8, use_graph=use_graph) as dev:
        dev.set_inducing(f["Xu"][:M]); dev.set_data(d["xtrain"], d["ytrain"]); dev.set_kernel(float(p[0]), p[1:], 0.0)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        dev.sweep(); ref = dev.posterior(); ref_sc = dev.scalars()
        t0 = time.perf_counter()
        for it in range(n_sweeps):
            dev.sweep()
            if it % 50 == 49:                 # results fetched while later sweeps are NOT yet queued, and ...
                cur = dev.posterior(); sc = dev.scalars()
                same = all(np.array_equal(a, b) for a, b in zip(cur, ref)) and sc == ref_sc
                bad += (not same)
                for _ in range(7): dev.sweep()   # ... bursts without any host synchronisation in between
        dt = time.perf_counter() - t0
    print(f"M={M} graph={use_graph}: {n_sweeps} sweeps, {bad} mismatches so far, {dt:.1f} s", flush=True)
sys.exit(1 if bad else 0)
