#!/usr/bin/env python3
"""Timeline of ONE sweep from in-kernel stamps (variant library built with -DSGP_SWEEP_TRACE): how the statistics streams, the
Lambda chain and the K_uu chain of an (overlapped) sweep really interleave -- without a profiler slowing the host's launches.
    python tools/sweep_trace.py [N M D] [sweeps]        (SGP_OVERLAP / SGP_OVERLAP_COLS as for the library; SGP_TRACE_SOLO=1: a fetch after every sweep)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocessnode_amd import _lib
from gaussianprocessnode_amd.device import SGPDevice
import bench

N, M, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (10000, 512, 8)
sweeps = int(sys.argv[4]) if len(sys.argv) >= 5 else 40
X, Xu, y, _, _ = bench.synthetic(N, M, D)
lib = _lib.load(variant="trace")
dev = SGPDevice.__new__(SGPDevice)
dev._lib = lib
dev._h = C.c_void_p()
cfg = _lib.Config(n_max=N, m=M, d=D, d_out=1, device=0, flags=0)
_lib.check(lib.sgp_create(C.byref(cfg), C.byref(dev._h)), None, "sgp_create", lib=lib)
dev.n_max, dev.M, dev.D, dev.d_out, dev.device, dev.Q, dev.n = N, M, D, 1, 0, M, 0
dev.set_inducing(Xu)
dev.set_data(X, y)
dev.set_kernel(bench.SIGMA2, bench.ELL[:D] if D <= len(bench.ELL) else np.full(D, 2.0), 0.0)
dev.set_prior_isotropic(bench.PRIOR_VAR)
dev.set_noise([[bench.W_BAR]])
print("plan:", dev.overlap_plan())
solo = os.environ.get("SGP_TRACE_SOLO") is not None      # one sweep at a time: the traced (last) sweep starts on an idle device
for _ in range(sweeps):
    dev.sweep()
    if solo:
        dev.scalars()
sc = dev.scalars()
buf = (C.c_int64 * (256 * 65))()
_lib.check(lib.sgp_get_sweep_trace(buf), None, "sgp_get_sweep_trace", lib=lib)
tr = np.array(buf[:], dtype=np.int64).reshape(256, 65)
names = {0: "prep_xu (stats)", 1: "prep_xu (K_uu)", 2: "gram_uf", 3: "gram_uu", 4: "trmv_mu_scan", 5: "gemm32 Sigma", 6: "gemm32 Kuu^-1",
         7: "scalars", 8: "join_wait (statM)", 9: "join_wait (K_uu chain's gate)", 10: "join_wait (statM: Gram done, B sum may start)"}
rows = []
for s in range(256):
    b, e = tr[s, 0], tr[s, 1:].max()
    if b == 0 or e == 0:
        if 200 <= s < 232 and b:
            rows.append((b, b, f"  step {s - 200} of the Lambda chain has its statistics"))
        continue
    if s in names: nm = names[s]
    elif 16 <= s < 40: nm = f"Lambda step {s - 16}"
    elif 40 <= s < 64: nm = f"  K_uu step {s - 40}"
    elif 64 <= s < 128: nm = f"syrk (first tile {s - 64})"
    elif 128 <= s < 192: nm = f"assemble (first tile row {s - 128})"
    else: nm = f"slot {s}"
    rows.append((b, e, nm))
t0 = min(r[0] for r in rows if r[2].startswith("prep_xu (stats)") or r[2] == "gram_uf")
for b, e, nm in sorted(rows):
    print(f"{(b - t0) / 100:9.1f} {(e - t0) / 100:9.1f} {(e - b) / 100:7.1f} us  {nm}")
if os.environ.get("SGP_TRACE_WGS"):
    # the chain steps have fewer than 64 workgroups: slot 1 + blockIdx of a step's record is that workgroup's own exit
    for s in list(range(16, 25)) + list(range(40, 49)):
        b = tr[s, 0]
        if not b: continue
        ends = [(int(e) - b) / 100 for e in tr[s, 1:] if e]
        print(("Lambda" if s < 40 else "K_uu") + f" step {s - 16 if s < 40 else s - 40}: workgroup exits after " + " ".join(f"{e:.1f}" for e in ends))
print(f"energy {sc.energy:.6f}")
