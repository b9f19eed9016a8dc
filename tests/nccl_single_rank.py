"""Helper of test_gpu_parity.py::test_torch_nccl_allreduce_inside_the_sweep_with_one_rank (launched through
torch.distributed.run with one rank): the production N > 1 path -- HipEngine + ShardedSweep, backend "nccl" (= RCCL), the
collective issued from the C ABI's all-reduce hook inside sgp_sweep on the sweep's stream -- with the one rank a one-GPU box has.
The reduce over one rank is the identity: results must be bitwise those of a handle without the hook."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from gaussianprocessnode_amd.distributed import HipEngine, ShardedSweep  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
rng = np.random.default_rng(4)
N, M, D = 5000, 200, 6
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = np.sin(X.sum(1))
res = []
for force in (False, True):
    eng = HipEngine(N, M, D, 1, device=0)
    dev = eng.dev
    dev.set_inducing(Xu); dev.set_data(X, y); dev.set_kernel(0.9, np.linspace(1.5, 3, D), 0.0)
    dev.set_prior_isotropic(50.0); dev.set_noise([[100.0]])
    sw = ShardedSweep(eng, force_hook=force)
    assert sw.hooked == force and sw.backend == ("nccl" if force else "none")
    for _ in range(5):
        sw.sweep()
    torch.cuda.synchronize()
    res.append((dev.posterior(), dev.scalars().energy))
    dev.close()
(mu0, S0, U0), e0 = res[0]
(mu1, S1, U1), e1 = res[1]
assert np.array_equal(mu0, mu1) and np.array_equal(S0, S1) and np.array_equal(U0, U1) and e0 == e1
dist.destroy_process_group()
print("nccl single-rank ok")
