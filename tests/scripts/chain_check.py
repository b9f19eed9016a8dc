#!/usr/bin/env python3
"""Quick check of the persistent factorisation launch through the stand-alone building blocks (sgp_potrf / sgp_potri)
against NumPy, for every tile count it supports, then one sweep against the oracle."""
import os, sys
os.environ.setdefault("SGP_CHAIN", "persistent"), time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd import device as Dv

rng = np.random.default_rng(3)
bad = 0
for n in (20, 64, 100, 128, 150, 192, 256, 300, 384, 512, 600, 700):
    B = rng.normal(size=(n, n)); A = B @ B.T / n + np.eye(n)
    t0 = time.perf_counter()
    L = Dv.potrf(A); Ai = Dv.potri(A)
    dt = time.perf_counter() - t0
    Lr = np.linalg.cholesky(A)
    eL = np.linalg.norm(L - Lr) / np.linalg.norm(Lr); eI = np.linalg.norm(Ai - np.linalg.inv(A)) / np.linalg.norm(np.linalg.inv(A))
    ok = eL < 1e-12 and eI < 1e-11
    bad += (not ok)
    print(f"n={n:4d} relerr L {eL:.2e} inv {eI:.2e} {'ok' if ok else 'FAIL'} ({dt*1e3:.1f} ms)", flush=True)
sys.exit(1 if bad else 0)
