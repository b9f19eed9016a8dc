#!/usr/bin/env python3
"""Stand-alone duration of the streaming SYRK (one launch, all tiles, all CUs) at workload T for chunk geometries given as
NC:ND pairs (SGP_SYRK_NC / SGP_SYRK_ND overrides of syrk_geometry), HIP events around 20 launches."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np, bench
    from gaussianprocessnode_amd import SGPDevice, _lib
    N, M, D = 10000, 512, 8
    X, Xu, y, _, _ = bench.synthetic(N, M, D)
    with SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(bench.SIGMA2, bench.ELL, 0.0)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        dev.sweep(); dev.scalars()
        us = [dev.time_kernel(_lib.SGP_T_SYRK, 20) for _ in range(3)]
        print(os.environ.get("SGP_SYRK_NC"), os.environ.get("SGP_SYRK_ND"), " ".join(f"{u:.2f}" for u in us), flush=True)
else:
    for pair in sys.argv[1:]:
        env = dict(os.environ, SGP_OVERLAP="0")
        if pair != "default":
            nc, nd = pair.split(":")
            env.update(SGP_SYRK_NC=nc, SGP_SYRK_ND=nd)
        subprocess.run([sys.executable, __file__, "one"], env=env)
