"""Host logic on the CPU: the reference's meta/helper mirrors, the counter == N product hook, statistics packing,
sharding, and the N > 1 path under gloo (world_size 2).  Compute goes through tests/cpu_engine.py (oracle-backed
test doubles); the product path itself is covered by the -m gpu tests."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from gaussianprocessnode_amd import meta as Mt
from gaussianprocessnode_amd import unisgp as U
from gaussianprocessnode_amd.distributed import pack_stats, padded, shard_bounds, stats_count, unpack_stats
from gaussianprocessnode_amd.distributions import (GammaShapeRate, MvNormalMeanCovariance, NormalMeanVariance,
                                                    PointMass, WishartFast)
from oracle import sgp_oracle as O
from tests.cpu_engine import OracleDevice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_meta_has_the_reference_field_order():
    """helper_functions/gp_helperfunction.jl:33-44: method, Xu, Psi0, Psi1_trans, Psi2, KuuL, kernel, Uv, counter, N."""
    names = [f for f in Mt.UniSGPMeta.__dataclass_fields__][:10]
    assert names == ["method", "Xu", "Psi0", "Psi1_trans", "Psi2", "KuuL", "kernel", "Uv", "counter", "N"]
    names = [f for f in Mt.MultiSGPMeta.__dataclass_fields__][:8]
    assert names == ["method", "Xu", "Psi0", "Psi1_trans", "Psi2", "Kuu_inverse", "kernel", "GPCache"]
    kern = Mt.SEARDKernel()
    m = Mt.UniSGPMeta(None, np.arange(1.0, 11.0), None, None, None, None, kern, None, 0, 10)
    assert Mt.getInducingInput(m).shape == (10, 1) and Mt.getKernel(m) is kern and Mt.getmethod(m) is None   # GPtest.jl:89-95


def test_helpers_match_the_reference_tests():
    """GPtest.jl:77-87,106-112"""
    rng = np.random.default_rng(0)
    A, B, D, a, b = rng.random((4, 4)), rng.random((3, 4)), rng.random((4, 4)), rng.random(4), rng.random(4)
    c = Mt.GPCache()
    assert np.array_equal(Mt.mul_A_B(c, B, A, 3, 4), B @ A)
    assert np.array_equal(Mt.mul_A_B(c, A, D, 4), A @ D)
    assert np.allclose(Mt.mul_A_B_A(c, A, D, 4), A @ D @ A)
    assert np.allclose(Mt.mul_A_B_At(c, B, A, 3, 4), B @ A @ B.T)
    assert np.allclose(Mt.mul_A_v(c, A, a, 4), A @ a)
    assert math.isclose(Mt.jdotavx(a, b), a @ b)
    blk = Mt.create_blockmatrix(A, 2, 2)
    assert np.array_equal(blk[1][0], A[2:4, 0:2]) and np.array_equal(blk[0][1], A[0:2, 2:4])


def test_metrics_and_batching():
    y = np.array([1.0, 2.0, 4.0, 7.0])
    assert math.isclose(Mt.SMSE(y, np.zeros(4)), np.mean(y * y) / np.var(y, ddof=1))
    assert Mt.num_error([1, 0, 1], [1, 1, 0]) == 2.0 and math.isclose(Mt.error_rate([1, 0, 1], [1, 1, 0]), 2 / 3)
    xb, yb = Mt.split2batch((list(range(11)), list(range(11))), 5)
    assert [len(b) for b in xb] == [5, 5, 1] and yb[2] == [10]
    k = Mt.SEARDKernel(softplus_params=True)
    s2, ell = k(np.array([0.0, 1.0]))
    assert math.isclose(s2, math.log(2.0)) and math.isclose(ell[0], math.log1p(math.e))


def test_gamma_product_and_wishart():
    g = GammaShapeRate(0.01, 0.5)
    for _ in range(4):
        g = g.prod(GammaShapeRate(1.5, 2.0))
    assert math.isclose(g.a, 0.01 + 4 * 0.5) and math.isclose(g.b, 0.5 + 8.0)       # a0 + N/2 (UniSGPnode.jl:215)
    w = WishartFast(4, np.diag([2.0, 4.0]))
    nu, V = w.params()
    assert nu == 4 and np.allclose(V, np.diag([0.5, 0.25]))                          # GPtest.jl:468-470


def test_counter_hook_runs_one_sweep_on_the_nth_message():
    """GPnode/UniSGPnode.jl:62-73: counter += 1 per product, marginal + meta.Uv refreshed when counter == N."""
    rng = np.random.default_rng(1)
    N, M = 12, 5
    Xu = np.linspace(-2, 2, M)
    X, y = rng.uniform(-2, 2, N), rng.normal(size=N)
    eng = OracleDevice(N, M, 1)
    meta = Mt.make_uni_meta(None, Xu, Mt.SEARDKernel(), N, engine=eng, jitter=1e-8)
    theta, w = PointMass(np.array([1.0, 1.0])), PointMass(25.0)
    prior = MvNormalMeanCovariance(np.zeros(M), 50.0 * np.eye(M))
    msgs = [U.rule_v(PointMass(y[i]), PointMass(X[i]), w, theta, meta) for i in range(N)]
    assert all(isinstance(m, U.BufferUniSGP) for m in msgs) and eng.calls == []        # O(1) tokens, no device work yet
    q = prior
    for i, m in enumerate(msgs):
        q = U.prod(q, m)
        if i < N - 1:
            assert isinstance(q, U.PendingMarginal) and meta.counter == i + 1
    assert isinstance(q, MvNormalMeanCovariance) and meta.counter == 0                  # :70
    assert [c[0] for c in eng.calls] == ["set_data", "sweep"]                           # exactly one sweep
    ref = O.vmp_sweep(Xu[:, None], X[:, None], y, None, 1.0, np.array([1.0]), 25.0, jitter=1e-8,
                      mu0=np.zeros(M), Sigma0=50.0 * np.eye(M))
    np.testing.assert_allclose(q.mean(), ref.mu_v, rtol=1e-10)
    np.testing.assert_allclose(meta.Uv, ref.Uv, rtol=1e-9, atol=1e-12)                  # :68-69
    # :w rule and average energy per point, through the same meta
    g = U.rule_w(PointMass(y[3]), PointMass(X[3]), q, theta, meta)
    I1, I2 = O.rule_w_point(X[3], y[3], 0.0, ref.mu_v, ref.Uv, ref.KuuL, Xu[:, None], 1.0, np.array([1.0]))
    assert g.shape() == 1.5 and math.isclose(g.rate(), 0.5 * (I1 + I2), rel_tol=1e-9)   # GPtest.jl:231-241
    Ue = U.average_energy(PointMass(y[3]), PointMass(X[3]), q, w, theta, meta)
    assert math.isclose(Ue, O.average_energy_point(I1, I2, 25.0, math.log(25.0)), rel_tol=1e-9)
    out = U.rule_out(PointMass(0.3), q, w, theta, meta)
    assert math.isclose(out.mean(), O.rule_out_point(0.3, ref.mu_v, 25.0, Xu[:, None], 1.0, np.array([1.0]))[0],
                        rel_tol=1e-10) and out.precision() == 25.0
    qw = U.rule_w_summed(meta, GammaShapeRate(0.01, 0.01))
    assert math.isclose(qw.a, 0.01 + N / 2) and math.isclose(qw.b, 0.01 + 0.5 * (ref.sum_I1 + ref.sum_I2), rel_tol=1e-9)


def test_wrong_N_and_mixed_parameters_are_errors():
    eng = OracleDevice(4, 3, 1)
    meta = Mt.make_uni_meta(None, [0.0, 1.0, 2.0], Mt.SEARDKernel(), 3, engine=eng)
    th = PointMass(np.array([1.0, 1.0]))
    U.rule_v(PointMass(0.1), PointMass(0.5), PointMass(2.0), th, meta)
    with pytest.raises(ValueError):
        U.rule_v(PointMass(0.1), PointMass(0.6), PointMass(3.0), th, meta)              # different q_w in one graph
    with pytest.raises((NotImplementedError, ValueError)):
        U.rule_v(PointMass(0.1), NormalMeanVariance(0.0, 1.0), PointMass(2.0), th, meta)   # mixed / no cubature rule
    # the :theta rule needs a cubature rule for an uncertain input (no CPU evaluation anywhere: the closure itself is
    # device-backed and covered by the GPU tests)
    with pytest.raises(ValueError):
        U.rule_theta(PointMass(0.1), NormalMeanVariance(0.0, 1.0), MvNormalMeanCovariance(np.zeros(3), np.eye(3)),
                     PointMass(2.0), meta)


def test_pack_unpack_and_shards():
    rng = np.random.default_rng(2)
    for m, d_out in [(5, 1), (64, 1), (70, 2)]:
        P = rng.random((m, m))
        P = P + P.T
        B = rng.random((m, d_out))
        Ryy = rng.random((d_out, d_out))
        buf = pack_stats(P, B, 1.5, 2.5, 3.0, Ryy if d_out > 1 else None)
        assert buf.size == stats_count(m, d_out) and padded(m) % 64 == 0
        P2, B2, s_yy, s_w, n, R2 = unpack_stats(buf, m, d_out)
        assert np.array_equal(P2, P) and np.array_equal(B2, B) and (s_yy, s_w, n) == (1.5, 2.5, 3.0)
        if d_out > 1:
            assert np.array_equal(R2, Ryy)
    for n, w in [(10, 3), (7, 8), (40000, 8), (0, 2)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np, torch, torch.distributed as dist
from gaussianprocessnode_amd.distributed import ShardedSweep, shard_bounds
from tests.cpu_engine import OracleShardEngine
from oracle import sgp_oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(4)
N, M, D = 203, 24, 2
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[:M].copy(); y = rng.normal(size=N)
s2, ell, w = 0.9, np.array([1.5, 2.0]), 40.0
lo, hi = shard_bounds(N, world, rank)
eng = OracleShardEngine(Xu, X[lo:hi], y[lo:hi], s2, ell, w, 50.0, 1e-8)
sw = ShardedSweep(eng)
assert sw.world == world and sw.hooked and sw.backend == "gloo"
sw.sweep(); sw.sweep()                      # the buffer is rebuilt every sweep (no accumulation across sweeps)
assert eng.hook_calls == 2                  # the exchange step ran INSIDE the engine's sweep (the C ABI's all-reduce hook)
ref = O.vmp_sweep(Xu, X, y, None, s2, ell, w, jitter=1e-8, Lambda0=np.eye(M) / 50.0, xi0=np.zeros(M))
assert np.linalg.norm(eng.res.mu_v - ref.mu_v) / np.linalg.norm(ref.mu_v) < 1e-10
assert np.linalg.norm(eng.res.Sigma_v - ref.Sigma_v) / np.linalg.norm(ref.Sigma_v) < 1e-10
assert abs(eng.res.sum_I2 - ref.sum_I2) < 1e-9 * abs(ref.sum_I2) and eng.res.stats.n == N
val, grad = sw.theta_objective()           # additive over shards: one more (tiny) all-reduce
f = lambda p: O.theta_objective(Xu, X, y, p[0], p[1:], ref.mu_v, ref.Uv, w, jitter=1e-8)
p0 = np.concatenate([[s2], ell])
g_ref = np.array([(f(p0 + 1e-6 * e) - f(p0 - 1e-6 * e)) / 2e-6 for e in np.eye(3)])
assert abs(val - f(p0)) < 1e-8 * abs(f(p0)) and np.allclose(grad, g_ref, rtol=1e-5, atol=1e-6)
print(f"rank {rank} ok", flush=True)
dist.destroy_process_group()
'''


def test_sharded_sweep_world_size_2_gloo(tmp_path):
    """N > 1 path: shard -> local statistics -> all-reduce(sum) -> replicated tail, two processes over gloo."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


TRAIN_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch
import torch.distributed as dist
from gaussianprocessnode_amd.distributed import ShardedDevice
from gaussianprocessnode_amd.train import perform_inference
from tests.cpu_engine import OracleDevice

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)
N, M, D, B = 96, 10, 2, 24
X = rng.uniform(-1.7, 1.7, (N, D)); Xu = X[rng.permutation(N)[:M]].copy(); y = np.sin(X.sum(1)) + 0.1 * rng.normal(size=N)
theta0 = np.array([0.3, 0.6, 0.2])

def reduce(a):                      # the hook's collective: every rank hands in its part, every rank gets the sum
    t = torch.from_numpy(a)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)

dev = OracleDevice(B, M, D)
dev.set_allreduce_array(reduce)
sharded = ShardedDevice(dev, rank, world)
qv, theta = perform_inference(theta0, X, y, Xu, sharded, batch_size=B, epochs=2, w_val=50.0, jitter=1e-8, device_paced=False)
# every rank swept only its slice of every minibatch ...
assert [c for c in dev.calls if c[0] == "set_data"][0][1] == B // world
# ... and all ranks end with the same theta and q(v)
both = [None] * world
dist.all_gather_object(both, (theta, qv.m))
assert all(np.array_equal(both[0][0], b[0]) and np.array_equal(both[0][1], b[1]) for b in both)
if rank == 0:
    one = OracleDevice(B, M, D)
    qv1, theta1 = perform_inference(theta0, X, y, Xu, one, batch_size=B, epochs=2, w_val=50.0, jitter=1e-8, device_paced=False)
    assert np.max(np.abs(theta - theta1)) < 1e-10 * np.max(np.abs(theta1)), (theta, theta1)
    assert np.linalg.norm(qv.m - qv1.m) < 1e-8 * np.linalg.norm(qv1.m)
    assert not np.array_equal(theta1, theta0)
print(f"rank {rank} ok", flush=True)
dist.destroy_process_group()
'''


def test_sharded_training_world_size_2_gloo(tmp_path):
    """The training loop of experiments/regression_kin40k.ipynb:196-230 data-sharded over two processes: every rank sweeps its
    slice of each minibatch, the statistics and the theta gradient are summed through the all-reduce hook, AdaMax runs
    replicated -- theta must come out as in the single-rank loop."""
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


# ------------------------------------------------------------------------------------------------
# the reference's toy experiments on its own saved data (tests/golden/toy*_fixture.npz), host loops over the CPU engine
# ------------------------------------------------------------------------------------------------
TOY_REG_THETA = np.array([0.03619750112503042, 0.5397814213749237])      # "Optimal hyperparameters", GPT_regression.ipynb cell 12
TOY_REG_SMSE = 0.008131895454357316                                      # "SMSE value of SGP node", cell 17
TOY_CLS_THETA = np.array([0.28307103179255905, 1.3847608644015856])      # GPT_classification.ipynb cell 11
TOY_CLS_ERRORS = 35                                                      # "Number of error:35.0", cell 21


def toy_fixture(kind):
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{kind}_fixture.npz"))
    return d["xtrain"], d["ytrain"], d["xtest"], d["ytest"], d["Xu"]


def test_toy_regression_reaches_the_notebooks_smse():
    """BASELINE config 1 on the reference's data: 7 VMP iterations with q(w) updates (GPT_regression.ipynb cells 6-9) at
    the printed optimum, prediction on the 600 test inputs (cells 14-15), SMSE against the printed 0.00813.  (The
    notebook's last q(v) was inferred one LBFGS round before the printed theta, so the match is to 1e-3, not to rounding.)"""
    from gaussianprocessnode_amd.meta import SMSE
    from gaussianprocessnode_amd.train import vmp_regression
    x, y, xt, yt, Xu = toy_fixture("toyregression")
    eng = OracleDevice(len(y), len(Xu), 1)
    qv, (a, b) = vmp_regression(TOY_REG_THETA, x, y, Xu, eng)
    assert a == 0.01 + 25.0 and 100.0 < a / b < 130.0                     # the data were drawn with precision 100 (cell 3)
    eng.set_kernel(TOY_REG_THETA[0], TOY_REG_THETA[1:], 1e-8)
    pred = eng.predict(xt.reshape(-1, 1), qv.m)
    assert abs(SMSE(yt, pred) - TOY_REG_SMSE) < 2e-3 * TOY_REG_SMSE


def test_toy_classification_reaches_the_notebooks_error_count():
    """GPT_classification.ipynb cells 7-9 (30 iterations of q(f), q(v), q(w)) at the printed optimum, then cells 13-21:
    the predicted class is 1 where the predictive mean of f is positive; the notebook counts 35 errors on its 400 test
    labels."""
    from gaussianprocessnode_amd.meta import num_error
    from gaussianprocessnode_amd.train import vmp_classification
    x, y, xt, yt, Xu = toy_fixture("toyclassification")
    eng = OracleDevice(len(y), len(Xu), 1)
    qv, (a, b) = vmp_classification(TOY_CLS_THETA, x, y, Xu, eng)
    pred = eng.predict(xt.reshape(-1, 1), qv.m)
    assert num_error(yt, (np.ravel(pred) > 0).astype(float)) == TOY_CLS_ERRORS
