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
# the training step through the same hook (statistics + the data half of the theta gradient, both reduced by RCCL inside
# sgp_train_step on the library's own stream): over one rank the reduce is the identity, theta must be that of a run without it
from gaussianprocessnode_amd.distributed import ShardedDevice  # noqa: E402
from gaussianprocessnode_amd.train import AdaMax, perform_inference  # noqa: E402
th0 = np.array([0.2, 0.9, 0.7, 1.1, 0.8, 1.0, 0.9])
out = []
for force in (False, True):
    eng = HipEngine(500, 64, D, 1, device=0)
    sw = ShardedSweep(eng, force_hook=force)
    qv, th = perform_inference(th0, X[:2000], y[:2000], Xu[:64], ShardedDevice(eng.dev, 0, 1), batch_size=500, epochs=2, w_val=100.0,
                               jitter=1e-8, optimizer=AdaMax(eta=0.01))
    torch.cuda.synchronize()
    out.append((th, qv.m))
    eng.dev.close()
# (not bitwise: with the hook the data half of the gradient is folded before the reduce, without it inside the finishing kernel)
assert np.allclose(out[0][0], out[1][0], rtol=1e-11, atol=0) and np.allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)
assert not np.array_equal(out[0][0], th0)
dist.destroy_process_group()
print("nccl single-rank ok")
