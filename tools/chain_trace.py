#!/usr/bin/env python3
"""Per-step timeline of the critical workgroup of the persistent factorisation launches (SGP_CHAIN_TRACE) at workload T."""
import os, sys, ctypes as C
os.environ["SGP_CHAIN_TRACE"] = "1"
os.environ["SGP_CHAIN"] = "persistent"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd import _lib
N, M, D = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "T"]
X, Xu, y, Xt, yt = bench.synthetic(N, M, D)
with G.SGPDevice(N, M, D) as dev:
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
    dev.set_prior_isotropic(bench.PRIOR_VAR); dev.set_noise([[bench.W_BAR]])
    for _ in range(20): dev.sweep()
    dev.posterior()
    ph, cnt = dev.phase_totals(reset=True)
    for _ in range(50): dev.sweep()
    dev.posterior()
    ph, cnt = dev.phase_totals()
    print("phases us:", {k: round(float(v), 1) for k, v in zip(("sweep", "gram", "syrk", "F1", "F2", "gap", "kuu", "local"), ph)}, "n", cnt)
    for which, name in ((1, "Lambda"), (0, "K_uu")):
        buf = (C.c_int64 * 384)()
        _lib.check(dev._lib.sgp_get_chain_trace(dev._h, which, buf), dev._h, "trace")
        t = np.array(buf, dtype=np.int64).reshape(12, 32)
        t0 = t[0, 0]
        if t[11, 31] > t[11, 29]:
            print(name, "chain: shader clock held by the critical workgroup: %.0f MHz" % ((t[11, 30] - t[11, 28]) / (t[11, 31] - t[11, 29]) * 100.0))
        print(name, "chain: step | begin | S-end | X wait | X arr | X solved | X tail | w0 tail | w1 tail | Dt recv || feeder: acc done | blk3 | sl3 wait | sl3 done | st w0 | st w4")
        for j in range((M + 63) // 64):
            r = [(t[j, k] - t0) / 100.0 if t[j, k] else float('nan') for k in (0, 1, 5, 2, 3, 4, 6, 7, 8, 13, 9, 14, 10, 11, 12)]
            print(f"  {j:2d} | " + " | ".join(f"{v:7.2f}" for v in r))
        print(name, "feeder wave 0 per column block: in regs | solved | published | updated")
        for j in range(1, (M + 63) // 64 - 1):
            print(f"  row {j+1}: " + "  ||  ".join(" ".join(f"{(t[j, 16 + 4 * cb + k] - t0) / 100.0:7.2f}" for k in range(4)) for cb in range(4)))
