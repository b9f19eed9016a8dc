"""Helper of test_gpu_parity.py::test_rccl_adapter_with_a_single_rank_communicator (run as a script in a child process):
a one-rank RCCL communicator made with ctypes, registered with sgp_use_rccl, must leave the sweep's results bitwise unchanged
(all-reduce over one rank = identity) while the all-reduce really runs inside sgp_sweep on the sweep's stream."""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: the process must hold ONE HIP runtime, torch's)
import gaussianprocessnode_amd as G  # noqa: E402

cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
rccl = C.CDLL(cands[0], mode=C.RTLD_GLOBAL)                 # RTLD_GLOBAL: sgp_use_rccl resolves ncclAllReduce with dlsym


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

rng = np.random.default_rng(3)
N, M, D = 3000, 130, 4
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = np.sin(X.sum(1))
out = []
for use in (False, True):
    with G.SGPDevice(N, M, D) as dev:
        dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.9, np.linspace(1.5, 3, D), 1e-8)
        dev.set_prior_isotropic(50.0); dev.set_noise([[20.0]])
        if use:
            dev.use_rccl(comm.value)
        for _ in range(3):
            dev.sweep()
        out.append((dev.posterior(), dev.scalars()))
(mu0, S0, U0), sc0 = out[0]
(mu1, S1, U1), sc1 = out[1]
assert np.array_equal(mu0, mu1) and np.array_equal(S0, S1) and np.array_equal(U0, U1) and sc0.energy == sc1.energy
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("rccl single-rank ok")
