#!/usr/bin/env python3
"""Per-phase timing of the Cholesky step kernel's critical workgroup (tile (j+1, j) of the Lambda chain), from a library built
with -DSGP_STEP_TRACE (the "trace" variant library):  python tools/step_trace.py      [SGP_OVERLAP=0 for the plain order]"""
import ctypes as C, os, sys
os.environ.setdefault("SGP_LIB_VARIANT", "trace")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussianprocessnode_amd as G
from gaussianprocessnode_amd import _lib

N, M, D = 10000, 512, 8
rng = np.random.default_rng(0)
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = np.sin(X.sum(1))
with G.SGPDevice(N, M, D) as dev:
    print("order:", "overlapped" if dev.__class__ and os.environ.get("SGP_OVERLAP") != "0" else "plain")
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.9, np.linspace(1.5, 3, D), 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
    for _ in range(20): dev.sweep()
    try:
        dev.scalars()
    except Exception as e:                      # (experimental builds may produce garbage; the stamps are still valid)
        print('note:', type(e).__name__)
    out = (C.c_int64 * 512)()
    _lib.load().sgp_get_step_trace(out)
t = np.array(out[:], dtype=np.int64).reshape(8, 2, 32)
names_f = ["entry", "tiles in LDS", "barrier", "potf2 done"]
# solve group (wave 0 of it): its work of interval I0 .. I7 done / the barrier behind it passed
names_x = ["entry", "tiles in LDS", "barrier", "I0", "b", "I1", "b", "I2", "b", "I3", "b", "solve done", "stored", "I4", "b", "I5", "b", "I6", "b",
           "I7", "b"]
for j in range(8):
    f, x = t[j, 0], t[j, 1]
    if f[0] == 0: continue
    t0 = min(f[0], x[0])
    print(f"step {j}: factoring " + " ".join(f"{n}={(f[i]-t0)/100:.2f}" for i, n in enumerate(names_f)))
    order = list(range(11)) + list(range(13, 21)) + [11, 12]
    print(f"        solve     " + " ".join(f"{names_x[i]}={(x[i]-t0)/100:.2f}" for i in order if x[i]))
    if j + 1 < 8 and t[j + 1, 0, 0]: print(f"        next step's entry at {(min(t[j+1,0,0], t[j+1,1,0]) - t0)/100:.2f}")
