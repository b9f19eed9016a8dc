"""Data-sharded VMP sweep: one process per GPU, points block-partitioned over the ranks, ONE
sum-all-reduce of the packed M x M statistics per sweep.

The reference has no counterpart (it is single-process); additivity of the statistics is what its
N-fold message product (GPnode/UniSGPnode.jl:62-63) and its sequential minibatch carry
(experiments/regression_kin40k.ipynb:205-212) already rely on.

`torch.distributed` is plumbing here: with backend "nccl" the all-reduce IS RCCL over xGMI.  The local
statistics and the replicated tail come from an *engine*; the product engine is `HipEngine` (the C ABI
on this rank's MI355X).  Tests inject a CPU engine to cover sharding, packing and the collective under
gloo -- there is no CPU fallback in the product: `HipEngine` raises without the HIP library / a GPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

TILE = 64
S_COUNT = 8          # SGP_S_COUNT of include/sgp_hip.h


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Block partition of n points: the first (n mod world) ranks get one extra point."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def padded(m: int) -> int:
    return (int(m) + TILE - 1) // TILE * TILE


def stats_count(m: int, d_out: int = 1) -> int:
    """Doubles in the packed statistics buffer [Psi2: Mp*Mp | B: Mp*d_out | scalars | Ryy: d_out^2]."""
    mp = padded(m)
    return mp * mp + mp * d_out + S_COUNT + d_out * d_out


def pack_stats(Psi2, B, s_yy: float, s_w: float, n_nodes: float, Ryy=None) -> np.ndarray:
    """Host-side packing in the device layout (column-major, padded) -- used by CPU test engines."""
    Psi2 = np.asarray(Psi2, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(Psi2.shape[0], -1)
    m, d_out = B.shape
    mp = padded(m)
    out = np.zeros(stats_count(m, d_out))
    P = np.zeros((mp, mp))
    P[:m, :m] = Psi2
    out[:mp * mp] = P.T.reshape(-1)                      # column-major
    Bp = np.zeros((d_out, mp))
    Bp[:, :m] = B.T
    out[mp * mp:mp * mp + mp * d_out] = Bp.reshape(-1)
    sc = out[mp * mp + mp * d_out:]
    sc[0], sc[1], sc[2] = s_yy, s_w, n_nodes
    if Ryy is not None:
        sc[S_COUNT:S_COUNT + d_out * d_out] = np.asarray(Ryy, dtype=np.float64).T.reshape(-1)
    elif d_out == 1:
        sc[S_COUNT] = s_yy
    return out


def unpack_stats(buf, m: int, d_out: int = 1):
    """Inverse of pack_stats: (Psi2 (M,M), B (M,d_out), s_yy, s_w, n_nodes, Ryy (d_out,d_out))."""
    buf = np.asarray(buf, dtype=np.float64)
    mp = padded(m)
    Psi2 = buf[:mp * mp].reshape(mp, mp).T[:m, :m].copy()
    B = buf[mp * mp:mp * mp + mp * d_out].reshape(d_out, mp)[:, :m].T.copy()
    sc = buf[mp * mp + mp * d_out:]
    Ryy = sc[S_COUNT:S_COUNT + d_out * d_out].reshape(d_out, d_out).T.copy()
    return Psi2, B, float(sc[0]), float(sc[1]), float(sc[2]), Ryy


def device_tensor(ptr: int, count: int, device="cuda"):
    """A float64 torch tensor over `count` doubles of device memory at `ptr` (no copy; __cuda_array_interface__) -- how a Python
    all-reduce hook looks at the buffer the library hands it."""
    import torch

    class _View:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}
    return torch.as_tensor(_View(), device=device)


class HipEngine:
    """This rank's shard on its MI355X: local statistics and the replicated tail through the C ABI."""

    def __init__(self, n_max: int, m: int, d: int, d_out: int = 1, device: int = 0, use_graph: bool = False):
        import torch
        from .device import SGPDevice
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a gfx950 GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(device)
        self.torch = torch
        self.dev = SGPDevice(n_max, m, d, d_out, device=device, use_graph=use_graph)
        self.stats = torch.zeros(stats_count(m, d_out), dtype=torch.float64, device=f"cuda:{device}")
        self.dev.bind_stats(self.stats.data_ptr())
        # One explicit (non-default) torch stream carries the sweep AND the collective: the C ABI treats a NULL stream as
        # "the library's own stream", which would not be ordered against torch's default stream or RCCL's.
        self.stream = torch.cuda.Stream(device=device)
        self._hooked = False
        torch.cuda.synchronize(device)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def sweep_local(self):
        self.dev.sweep_local(self.stream.cuda_stream)

    def sweep_finish(self):
        self.dev.sweep_finish(self.stream.cuda_stream)

    def device_view(self, ptr: int, count: int):
        if ptr == self.stats.data_ptr() and count == self.stats.numel():
            return self.stats
        return device_tensor(ptr, count, self.stats.device)

    def install_allreduce(self, reduce_tensor):
        """Make the exchange step part of the library's calls (include/sgp_hip.h, sgp_set_allreduce): `reduce_tensor(t)` must
        sum-all-reduce the torch tensor `t` in place on the CURRENT torch stream.  The library hands the hook a device buffer
        and the stream it wants the collective on: the packed statistics inside `sgp_sweep` / `sgp_train_step`, the data
        half of the theta gradient (33 doubles) inside `sgp_theta_objective` / `sgp_train_step`."""
        torch = self.torch

        def hook(buf, count, stream):
            t = self.device_view(buf, count)
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=t.device)) if stream else self.stream_context()
            with ctx:
                reduce_tensor(t)
        self.dev.set_allreduce(hook)
        self._hooked = True

    def sweep(self):
        """local statistics -> installed all-reduce hook -> replicated tail: ONE library call on the library's own streams (NULL
        stream), where it may overlap the statistics with the Lambda chain (include/sgp_hip.h, sgp_overlap_plan) -- hooked or not:
        the hook is handed the stream each piece of the exchange buffer must be reduced on (one piece per statistics group in the
        overlapped order) and enters it, so the collective is ordered where the library needs it; the getters wait for the sweep."""
        self.dev.sweep(0)

    def synchronize(self):
        self.torch.cuda.synchronize()

    reduces_theta_objective = True       # with the hook installed sgp_theta_objective returns the sum over all shards

    def theta_objective_local(self, n_ell=None):
        """The hyper-parameter objective and its gradient at the last sweep's q(v): with the all-reduce hook installed the
        value and the gradient of ALL shards (summed inside the library), without it this shard's alone."""
        return self.dev.theta_objective(want_grad=True, n_ell=n_ell)


class ShardedSweep:
    """local statistics -> all-reduce(sum) -> replicated tail.  `engine` provides sweep_local(),
    sweep_finish() and a torch tensor `stats`; `group` is a torch.distributed process group (None = default;
    no collective is issued when torch.distributed is not initialised or the world has one rank)."""

    def __init__(self, engine, group=None, force_hook=False):
        self.engine = engine
        self.group = group
        import torch.distributed as dist
        self.dist = dist
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        # force_hook: install the collective even in a one-rank world (what a one-GPU box can rehearse: the RCCL
        # all-reduce of torch.distributed issued from inside sgp_sweep on the sweep's stream)
        force_hook = bool(force_hook) and live
        self.backend = dist.get_backend(group) if (self.world > 1 or force_hook) else "none"
        # Engines that run the exchange step inside their own sweep (HipEngine: the C ABI's all-reduce hook) get the
        # collective installed once; the others are driven half by half.
        self.hooked = (self.world > 1 or force_hook) and hasattr(engine, "install_allreduce")
        if self.hooked:
            engine.install_allreduce(self._reduce_tensor)

    def _reduce_tensor(self, t):
        """The collective of the library's hook: sum `t` over the ranks in place, on the current stream (HipEngine enters the
        stream the library named before it calls this)."""
        if self.backend == "gloo" and t.is_cuda:
            # (rehearsal on a one-GPU box: gloo carries host tensors)
            import torch
            torch.cuda.current_stream().synchronize()
            host = t.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _reduce(self):
        # engines that are driven half by half (sweep_local / sweep_finish): the statistics tensor, inside the engine's stream
        # context -- the collective is ordered after the local kernels and before the replicated tail on that stream
        ctx = self.engine.stream_context() if hasattr(self.engine, "stream_context") else None
        if ctx is not None:
            ctx.__enter__()
        try:
            self._reduce_tensor(self.engine.stats)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def sweep(self):
        if self.hooked or (self.world == 1 and hasattr(self.engine, "sweep")):
            self.engine.sweep()
            return
        self.engine.sweep_local()
        if self.world > 1:
            self._reduce()
        self.engine.sweep_finish()

    def theta_objective(self, n_ell=None):
        """neg_log_backwardmess_fast and its gradient over ALL shards (helper_functions/derivative_helper.jl:23-39,55-63):
        each rank contributes its points' terms at the replicated q(v).  With the library's hook installed the sum happens
        inside `sgp_theta_objective` (the data half of the gradient through the hook; value, K_uu half and s_w term from the
        reduced statistics), nothing is recomputed and every rank gets the whole result; engines without it add one small
        all-reduce here."""
        import numpy as np
        value, grad = self.engine.theta_objective_local(n_ell)
        if self.hooked and getattr(self.engine, "reduces_theta_objective", False):
            return float(value), np.asarray(grad, dtype=np.float64).copy()
        packed = np.concatenate([[value], np.asarray(grad, dtype=np.float64)])
        if self.world > 1:
            import torch
            t = torch.from_numpy(packed)
            if self.dist.get_backend(self.group) == "nccl":
                t = t.to(self.engine.stats.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            packed = t.cpu().numpy()
        return float(packed[0]), packed[1:].copy()


class ShardedDevice:
    """One rank of a data-sharded run behind the `SGPDevice` interface the drivers use (`train.perform_inference`, host- or
    device-paced): the caller hands every rank the SAME minibatches; this wrapper keeps the rank's slice of each and lets the
    library sum statistics and gradients over the ranks through its all-reduce hook, so that every rank ends a step with the
    same q(v) and the same theta.  The reference has no counterpart (single process); the additivity it relies on is the
    N-fold product of GPnode/UniSGPnode.jl:62-63 and the sum over points of helper_functions/derivative_helper.jl:29-38.

    `dev`: the rank's device (an `SGPDevice`, or a test double with the same methods); `rank`, `world`: the partition.
    The hook is the caller's business (`HipEngine.install_allreduce` / `ShardedSweep` for torch.distributed, `sgp_use_rccl`
    for a bare communicator, a test double)."""

    def __init__(self, dev, rank: int, world: int):
        self.dev, self.rank, self.world = dev, int(rank), int(world)

    def __getattr__(self, name):            # everything that is replicated (setters of theta, prior, noise; getters) passes through
        return getattr(self.dev, name)

    def _slice(self, n):
        return shard_bounds(n, self.world, self.rank)

    def set_data(self, X, y_mean, y_var=None, weights=None, n_nodes=None):
        X = np.asarray(X, dtype=np.float64)
        n = len(X) if X.ndim > 1 else len(np.atleast_1d(y_mean))
        lo, hi = self._slice(n)
        cut = lambda a: None if a is None else np.asarray(a)[lo:hi]
        nn = None if n_nodes is None else float(n_nodes) * (hi - lo) / max(n, 1)
        self.dev.set_data(X.reshape(n, -1)[lo:hi], cut(y_mean), cut(y_var), cut(weights), nn)

    def train_step(self, offset, n, learn=True, reset_prior=False):
        lo, hi = self._slice(n)
        self.dev.train_step(offset + lo, hi - lo, learn, reset_prior)
