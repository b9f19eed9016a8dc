#!/usr/bin/env python3
"""Rehearsal of the N > 1 path on a ONE-GPU box: two ranks share device 0, gloo carries the all-reduce (RCCL refuses two
ranks on one device).  Checks the sharded posterior against the single-rank posterior.
Launch:  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/rehearse_two_ranks.py"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
def say(msg):
    print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

say("init gloo")
dist.init_process_group(backend="gloo")
from gaussianprocessnode_amd.distributed import HipEngine, ShardedSweep, shard_bounds
import gaussianprocessnode_amd as G
# (RH_N=20000 RH_M=512 RH_D=8 RH_JIT=1e-6: every rank's shard then qualifies for the OVERLAPPED order -- one reduce per statistics
# group, on two streams; the default size runs the plain order)
N, M, D = int(os.environ.get("RH_N", 3000)), int(os.environ.get("RH_M", 128)), int(os.environ.get("RH_D", 4))
JIT, GTOL = float(os.environ.get("RH_JIT", 1e-8)), float(os.environ.get("RH_GTOL", 1e-6))
rng = np.random.default_rng(0)
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = np.sin(X.sum(1))
lo, hi = shard_bounds(N, world, rank)
say(f"engine for points [{lo}, {hi})")

eng = HipEngine(hi - lo, M, D, 1, device=0)
dev = eng.dev
dev.set_inducing(Xu); dev.set_data(X[lo:hi], y[lo:hi]); dev.set_kernel(0.9, np.full(D, 1.5), JIT)
dev.set_prior_isotropic(50.0); dev.set_noise([[100.0]])
sw = ShardedSweep(eng)
say(f"sweep order: {'overlapped, groups ' + str([(g['col_begin'], g['col_end']) for g in dev.overlap_plan()]) if dev.overlap_plan() else 'plain'}")
for it in range(3):
    say(f"sweep {it}")
    sw.sweep()
    eng.synchronize()
mu, Sig, _ = dev.posterior(want_uv=False)
val_s, grad_s = sw.theta_objective(n_ell=D)
say("single-rank reference")
with G.SGPDevice(N, M, D) as ref:
    ref.set_inducing(Xu); ref.set_data(X, y); ref.set_kernel(0.9, np.full(D, 1.5), JIT)
    ref.set_prior_isotropic(50.0); ref.set_noise([[100.0]]); ref.sweep()
    mu1, Sig1, _ = ref.posterior(want_uv=False)
    val_1, grad_1 = ref.theta_objective(want_grad=True, n_ell=D)
err = np.linalg.norm(mu - mu1) / np.linalg.norm(mu1), np.linalg.norm(Sig - Sig1) / np.linalg.norm(Sig1)
say(f"sharded vs single-rank: mu {err[0]:.2e} Sigma {err[1]:.2e}")
gerr = abs(val_s - val_1) / abs(val_1), np.linalg.norm(grad_s - grad_1) / np.linalg.norm(grad_1)
say(f"sharded theta objective / gradient vs single-rank: {gerr[0]:.2e} {gerr[1]:.2e}")
assert max(gerr) < GTOL
assert max(err) < 1e-7          # the two halves of Psi2 are summed in a different order: cond(Lambda) * eps
# the training loop sharded the same way (experiments/regression_kin40k.ipynb:196-230): every rank sweeps its slice of each
# minibatch; statistics and the data half of the theta gradient go through the hook inside sgp_train_step; AdaMax replicated
from gaussianprocessnode_amd.distributed import ShardedDevice
from gaussianprocessnode_amd.train import AdaMax, perform_inference
th0 = np.array([0.2, 0.9, 0.7, 1.1, 0.8, 1.0, 0.6, 1.2, 0.9])[:D + 1]
bs = 600
eng_t = HipEngine(bs, M, D, 1, device=0)
sw_t = ShardedSweep(eng_t)                       # installs the gloo all-reduce as the library's hook
qv, th = perform_inference(th0, X, y, Xu, ShardedDevice(eng_t.dev, rank, world), batch_size=bs, epochs=2, w_val=100.0, jitter=1e-8,
                           optimizer=AdaMax(eta=0.01))
eng_t.synchronize()
with G.SGPDevice(bs, M, D) as ref:
    qv1, th1 = perform_inference(th0, X, y, Xu, ref, batch_size=bs, epochs=2, w_val=100.0, jitter=1e-8, optimizer=AdaMax(eta=0.01))
terr = np.max(np.abs(th - th1)) / np.max(np.abs(th1)), np.linalg.norm(qv.m - qv1.m) / np.linalg.norm(qv1.m)
say(f"sharded training (device-paced, {2 * (N // bs)} steps) vs single rank: theta {terr[0]:.2e}, mu_v {terr[1]:.2e}")
assert terr[0] < 1e-9 and terr[1] < 1e-6
# classification (Probit, q(w) carried on the device): shape and rate of q(w) must count the whole minibatch on every rank
from gaussianprocessnode_amd.train import perform_inference_classification
lab = (y > 0).astype(np.float64)
thc = np.log(np.expm1(np.ones(D + 1)))
eng_c = HipEngine(bs, M, D, 1, device=0)
sw_c = ShardedSweep(eng_c)
qc, abc, thc_s = perform_inference_classification(thc, X, lab, Xu, ShardedDevice(eng_c.dev, rank, world), batch_size=bs, epochs=1,
                                                  optimizer=AdaMax(), device_paced=True)
eng_c.synchronize()
with G.SGPDevice(bs, M, D) as ref:
    qc1, abc1, thc_1 = perform_inference_classification(thc, X, lab, Xu, ref, batch_size=bs, epochs=1, optimizer=AdaMax(), device_paced=True)
cerr = np.max(np.abs(thc_s - thc_1)) / np.max(np.abs(thc_1)), abs(abc[1] - abc1[1]) / abs(abc1[1])
say(f"sharded Probit training ({N // bs} steps) vs single rank: theta {cerr[0]:.2e}, q(w) shape {abc[0]} vs {abc1[0]}, rate {cerr[1]:.2e}")
assert abc[0] == abc1[0] and cerr[0] < 1e-7 and cerr[1] < 1e-7
dist.barrier()
dist.destroy_process_group()
say("ok")
