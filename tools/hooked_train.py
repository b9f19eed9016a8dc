#!/usr/bin/env python3
"""A data-sharded device-paced training run on one GPU (the all-reduce hook doubles what it is handed: a second rank with the same
slice), 40 minibatches of 500 kin40k-shaped points, M = 512 -- run under rocprofv3 --kernel-trace --stats to count launches per
minibatch: one k_gram_uf and one SYRK launch each (nothing is re-formed for the theta gradient)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaussianprocessnode_amd.distributed import HipEngine, ShardedDevice
from gaussianprocessnode_amd.train import AdaMax, perform_inference

N, M, D, bs = 10000, 512, 8, 500
X, Xu, y, _, _ = bench.synthetic(N, M, D)
eng = HipEngine(bs, M, D, 1, device=0)
calls = []
def double(t):
    calls.append(t.numel())
    t.mul_(2.0)
eng.install_allreduce(double)
t0 = time.perf_counter()
qv, th = perform_inference(np.log(np.expm1(np.ones(D + 1))), X, y, Xu, ShardedDevice(eng.dev, 0, 1), batch_size=bs, epochs=2, w_val=1e4,
                           optimizer=AdaMax())
torch.cuda.synchronize()
eng.dev.close()
print(json.dumps({"minibatches": 2 * (N // bs), "seconds": time.perf_counter() - t0, "hook_calls": len(calls),
                  "hook_payload_doubles": sorted(set(calls)), "theta": [float(v) for v in th]}))
