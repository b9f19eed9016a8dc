#!/usr/bin/env python3
"""bench.py -- VMP iterations/sec of the sparse-GP node's sweep on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload T|C2|C3|N1M]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full VMP sweep over the resident synthetic data set: K_uu + Cholesky,
K_uf, the summed :v messages (Psi2, b), q(v) incl. Sigma_v and Uv, the summed :w statistics and the summed
average energy.  Inputs are resident in HBM before the timed region.  With N > 1 the points are sharded
over the ranks (strong scaling: the data set is fixed) and the packed statistics are summed by one RCCL
all-reduce per sweep.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, M, D)   -- BASELINE.json configs; T is the configuration the metric is quoted on
    "T": (10000, 512, 8),
    "C2": (10000, 256, 8),
    "C3": (40000, 512, 8),
    # scaling workload: at T the replicated M^3 tail caps strong scaling at ~1.35x on 8 GPUs; the >= 6x
    # regime of the north star needs the data-sized kernels to dominate
    "N1M": (1000000, 512, 8),
    "N100K": (100000, 512, 8),         # (diagnostic: between the latency-bound and the data-bound regime)
    "N5K": (5000, 512, 8),             # (diagnostic: one rank's shard of T on two GPUs)
}
# trained kin40k hyper-parameters (softplus(theta_opt), experiments/regression_kin40k.ipynb:255-263)
SIGMA2 = 0.17636613718898136
ELL = np.array([2.994391934274809, 2.905302600576806, 1.7401945529137626, 2.2697267449222425,
                2.0114338358466854, 1.5824668119572332, 1.533898096437981, 2.052099122165972])
W_BAR = 1e4            # experiments/regression_kin40k.ipynb:118
PRIOR_VAR = 50.0       # :203-204
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD spec; tools/mfma_f64_probe.hip saturates at ~49)
PIVOT_CYCLES = 127             # dependent cycles per pivot of the factorisation's inner loop: the chains' floor


def pmc_traffic(workload, world):
    """HBM/fabric bytes per SYRK launch from this round's rocprofv3 --pmc passes (tools/measure_round.sh writes the file next to
    the raw summaries it was computed from; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane reads)."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if rec.get("workload") != workload or rec.get("n_gpus") != world:
        return None, None
    return rec.get("traffic_bytes_per_launch"), {k: rec.get(k) for k in ("fetch_size_bytes", "write_size_bytes", "commit", "source")}


def synthetic(N, M, D, seed=1, n_test=2000):
    """X ~ U(-1.745, 1.745), Xu = first M rows of a seeded permutation, y = sin(sum x) + 0.1 eps,
    standardised."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1.745, 1.745, (N, D))
    Xu = X[rng.permutation(N)[:M]].copy()
    f = lambda Z: np.sin(Z.sum(axis=1))
    y = f(X) + 0.1 * rng.normal(size=N)
    mean, std = y.mean(), y.std()
    y = (y - mean) / std
    Xt = rng.uniform(-1.745, 1.745, (n_test, D))
    yt = (f(Xt) + 0.1 * rng.normal(size=n_test) - mean) / std
    return X, Xu, y, Xt, yt


def _rate(dev, reps, per_point=False, blocks=5):
    """Back-to-back sweeps per second on a prepared device (with the per-point :w quantities fetched every sweep if asked):
    the MEDIAN of `blocks` timed blocks of reps / blocks sweeps, like the headline -- on the shared boxes the host thread loses a
    10 ms scheduler tick a few times per thousand iterations (tools/per_point_outliers.py), a fifth of a 200-sweep measurement."""
    for _ in range(10):
        dev.sweep()
        if per_point:
            dev.w_stats()
    dev.scalars()
    per = max(1, reps // blocks)
    rates = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(per):
            dev.sweep()
            if per_point:
                dev.w_stats()
        dev.scalars()
        rates.append(per / (time.perf_counter() - t0))
    return float(np.median(rates))


def extras(headline):
    """After the timed region (rank 0, one GPU; a few tens of milliseconds each):
    `configs`   sweeps/s of every BASELINE.json configuration on this box (tools/config_rates.py's shapes);
    `per_point` what the drop-in calls through the real plugin API when q(w) is random: every :w message needs its own
                I1_n / I2_n (GPnode/UniSGPnode.jl:196-238), i.e. sgp_w_stats after every sweep -- the rate with that call in
                the loop, and the roofline entry of its dominant kernel k_quadform_fused."""
    from gaussianprocessnode_amd import SGPDevice, _lib
    shapes = [("C1 toy regression (GPT_regression.ipynb)", 50, 20, 1, 1), ("C2 kin40k M=256", 10000, 256, 8, 1),
              ("T kin40k M=512", 10000, 512, 8, 1), ("C3 kin40k N=40000 on one GPU", 40000, 512, 8, 1),
              ("C4 banana shape", 4000, 128, 2, 1), ("C5 pendulum MultiSGP (300 steps x 5 cubature points)", 1500, 48, 2, 2)]
    out = {"configs": [], "per_point": []}
    for name, N, M, D, Do in shapes:
        rng = np.random.default_rng(0)
        X = rng.uniform(-1.7, 1.7, (N, D))
        Xu = rng.uniform(-1.7, 1.7, (M, D))
        Y = np.sin(X.sum(1))[:, None] * np.ones((1, Do))
        per_point = Do == 1 and (name.startswith("T ") or name.startswith("C4"))
        with SGPDevice(N, M, D, d_out=Do, keep_kuf=per_point) as dev:
            dev.set_inducing(Xu)
            dev.set_data(X, Y if Do > 1 else Y[:, 0], None, np.full(N, 0.2) if Do > 1 else None, n_nodes=(N // 5 if Do > 1 else None))
            dev.set_kernel(1.0, np.full(D, 1.5), 1e-6)
            dev.set_prior_isotropic(50.0)
            dev.set_noise(np.eye(Do) * 10.0)
            reps = 200 if N <= 10000 else 60
            rate = _rate(dev, reps)
            out["configs"].append({"config": name, "N": N, "M": M, "D": D, "d_out": Do, "sweeps_per_s": rate,
                                   "order": "overlapped" if dev.overlap_plan() else "plain"})
            if per_point:
                rate_pp = _rate(dev, reps, per_point=True)
                q_us = dev.time_kernel(_lib.SGP_TIME_QUADFORM, 10)
                flops = 2.0 * float(N) * M * (M + 64)            # two lower-triangular factors, 64-wide tiles incl. the diagonal ones
                tf = flops / (q_us * 1e-6) / 1e12
                out["per_point"].append({
                    "config": name, "sweeps_per_s_with_w_stats": rate_pp, "sweeps_per_s_without": rate,
                    "what": "sgp_sweep + sgp_w_stats (k_quadform_fused + k_w_point_finish, then 2 n doubles to the host in one copy) per iteration; median of 5 blocks of 40",
                    "roofline_quadform": {"kernel": "k_quadform_fused (|L^-1 k_n|^2, |Uv k_n|^2 and k_n . mu in one pass over the resident K_uf)", "bound": "mfma",
                                          "launch_us": q_us, "algorithmic_flops_per_launch": flops, "achieved": tf,
                                          "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS}})
    return out


def scaling_leg(world, rank, local_rank, barrier):
    """N > 1 only (or SGP_BENCH_SCALING=1): the same sweep on the N1M workload (N = 10^6, M = 512), sharded the same way, right
    after the headline workload.  At T the replicated M^3 tail is two thirds of the sweep and no number of GPUs helps (DESIGN.md
    section 5: 1.05x / 1.05x / 0.91x predicted); this is the regime the design does scale in (6.6x predicted at 8 GPUs), so one
    SCALE run records both.  Returns sweeps/s (median of 3 blocks of 10 sweeps, max over ranks) and rank 0's phase durations."""
    import torch
    import torch.distributed as dist
    from gaussianprocessnode_amd import _lib
    from gaussianprocessnode_amd.distributed import HipEngine, ShardedSweep, shard_bounds
    N, M, D = WORKLOADS["N1M"]
    X, Xu, y, _, _ = synthetic(N, M, D, n_test=1)
    lo, hi = shard_bounds(N, world, rank)
    eng = HipEngine(hi - lo, M, D, 1, device=local_rank)
    dev = eng.dev
    dev.set_inducing(Xu)
    dev.set_data(X[lo:hi], y[lo:hi])
    dev.set_kernel(SIGMA2, ELL, 0.0)
    dev.set_prior_isotropic(PRIOR_VAR)
    dev.set_noise([[W_BAR]])
    sw = ShardedSweep(eng)
    for _ in range(3):
        sw.sweep()
    dev.wait(); barrier()
    dev.phase_totals(reset=True)
    steps, block_s = 10, []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sw.sweep()
        dev.wait(); barrier()
        block_s.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(block_s, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in t.cpu()]
    phase_us, _ = dev.phase_totals()
    sc = dev.scalars()
    dev.close()
    med = float(np.median(block_s))
    return {"workload": f"N1M: N={N} M={M} D={D}, the same sweep and hyper-parameters, data-sharded x{world}",
            "sweeps_per_s": steps / med, "ms_per_sweep": 1e3 * med / steps, "points_per_gpu": hi - lo, "scaling": "strong",
            "blocks_ms_per_sweep": [1e3 * b / steps for b in block_s], "energy": sc.energy,
            "phases_us_rank0": {"sweep_device": float(phase_us[_lib.SGP_T_SWEEP]), "local_statistics": float(phase_us[_lib.SGP_T_LOCAL]),
                                "gap_local_to_finish_incl_allreduce": float(phase_us[_lib.SGP_T_GAP_LOCAL_FINISH]),
                                "finish1_lambda_chain": float(phase_us[_lib.SGP_T_FINISH1]),
                                "finish2_traces": float(phase_us[_lib.SGP_T_FINISH2])},
            "collective": {"backend": sw.backend, "payload_doubles": int(eng.stats.numel())}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="T", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=None,
                    help="back-to-back timed blocks of --steps sweeps each (default 15, 5 at N > 1e5); the MEDIAN block is reported")
    args = ap.parse_args()

    # Host side of this rank on the CPUs of the GPU's NUMA node, before torch or HIP create a thread (what numactl --cpunodebind
    # does; gaussianprocessnode_amd/hostbind.py says why: an unconfined process on a two-socket host runs its short blocks 6 % slower
    # one start in four).  Loaded by path: importing the package would load the HIP runtime first.
    import importlib.util
    _spec = importlib.util.spec_from_file_location("_sgp_hostbind", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                 "gaussianprocessnode_amd", "hostbind.py"))
    hostbind = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(hostbind)
    host_binding = hostbind.bind_to_gpu_node(0 if os.environ.get("SGP_BENCH_REHEARSAL") is not None else int(os.environ.get("LOCAL_RANK", "0")))

    import torch
    import torch.distributed as dist
    from gaussianprocessnode_amd import _lib
    from gaussianprocessnode_amd.distributed import HipEngine, ShardedSweep, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: SGP_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo for the collective
    # (RCCL refuses two ranks on one device); numbers from such a run mean nothing, it only exercises the N > 1 code path
    rehearsal = os.environ.get("SGP_BENCH_REHEARSAL") is not None
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if host_binding["how"] != "none":
        # check the guess against the runtime's own answer (enumeration orders can differ); correct every thread if it was wrong
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
            host_binding["runtime_bdf"] = bdf
            if node >= 0 and node != host_binding["node"] and hostbind.rebind_all_threads(node):
                host_binding.update(node=node, how="gpu (corrected after the runtime started)")
        except (OSError, AttributeError, ValueError):
            pass
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))

    N, M, D = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = 300 if N <= 100000 else 20
    if args.warmup is None:
        args.warmup = 30 if N <= 100000 else 3
    X, Xu, y, Xt, yt = synthetic(N, M, D)
    lo, hi = shard_bounds(N, world, rank)

    eng = HipEngine(hi - lo, M, D, 1, device=local_rank, use_graph=os.environ.get("SGP_BENCH_GRAPH") is not None)
    dev = eng.dev
    dev.set_inducing(Xu)
    dev.set_data(X[lo:hi], y[lo:hi])
    dev.set_kernel(SIGMA2, ELL, 0.0)
    dev.set_prior_isotropic(PRIOR_VAR)
    dev.set_noise([[W_BAR]])
    sweep = ShardedSweep(eng)

    def barrier():
        # (sgp_wait first: it POLLS the library's streams, so the blocking synchronize behind it returns at once -- a blocking wait
        # alone may sleep until an interrupt, tens of microseconds that are the host's, not the sweep's, and that weigh on a block
        # of 20 sweeps.  The bracket the contract asks for -- synchronize, barrier, synchronize -- is unchanged.)
        dev.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: R back-to-back blocks, each EXACTLY --steps full sweeps bracketed by barrier + synchronize on both sides (max
    # over ranks per block); the MEDIAN block is the reported one.  One block of 20 sweeps is 5 ms of timing -- a single hiccup of
    # the host thread moved round 3's driver line by 3 % against the same process' 200-sweep figure.
    if args.blocks is None:
        args.blocks = 15 if N <= 100000 else 5
    for _ in range(args.warmup):
        sweep.sweep()
    barrier()
    dev.phase_totals(reset=True)
    block_s = []
    for _ in range(max(1, args.blocks)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sweep.sweep()
        barrier()
        block_s.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(block_s, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in t.cpu()]
    elapsed = float(np.median(block_s))

    # ---- live per-kernel numbers (this rank's shard) ------------------------------------------------
    # (1) averages over the K timed sweeps, from in-kernel first-block-in / last-block-out stamps (100 MHz clock)
    #     accumulated on the device inside the timed sweeps -- no host interaction in the timed region;
    # (2) HIP events around eager launches of the same kernel on the same stream, right after the timed region (the
    #     kernel alone on the chip, launch overhead included).
    phase_us, n_counted = dev.phase_totals()
    tick_us = lambda i: float(phase_us[i])
    stream = eng.stream.cuda_stream
    n_loc = hi - lo
    syrk_flops = float(n_loc) * M * (M + 1)                     # SYRK lower half n M (M+1)
    # The SYRK of the timed sweeps.  Plain order: ONE launch over all lower tiles.  Overlapped sweep (sgp_overlap_plan): one launch
    # per statistics group, the first on all CUs, the others on the CU-masked stream -- each is timed alone with HIP events on
    # the stream (and the CUs) it uses inside the sweep; `achieved` = the sweep's SYRK flops / the sum of its launches =
    # (flops per launch) / (average launch duration), the quantity rocprofv3 --kernel-trace --stats shows for k_syrk_direct.
    plan = dev.overlap_plan()            # (hooked sweeps too: one all-reduce per statistics group, include/sgp_hip.h)
    ntiles_all = (M + 63) // 64 * ((M + 63) // 64 + 1) // 2
    # (SGP_BENCH_SKIP_ALONE: the rocprofv3 --pmc passes of tools/measure_round.sh -- every k_syrk_direct launch in their output is
    # then one of the timed sweeps' own; the stand-alone timings are skipped and the roofline block is not meaningful)
    skip_alone = os.environ.get("SGP_BENCH_SKIP_ALONE") is not None
    if skip_alone:
        syrk_groups, syrk_launches, syrk_us_alone = None, max(len(plan), 1), float("nan")
    elif plan:
        group_us = [dev.time_group(g, 10) for g in range(len(plan))]
        syrk_groups = [{"tile_columns": [g["col_begin"], g["col_end"]], "tiles": g["tiles"], "point_chunks": g["chunks"], "cus": g["cus"],
                        "launch_us": us, "tflops": syrk_flops * g["tiles"] / ntiles_all / (us * 1e-6) / 1e12,
                        "frac_of_peak_of_its_cus": syrk_flops * g["tiles"] / ntiles_all / (us * 1e-6) / 1e12
                                                   / (FP64_MFMA_PEAK_TFLOPS * g["cus"] / 256.0)}
                       for g, us in zip(plan, group_us)]
        syrk_launches = len(plan)
        syrk_us_alone = sum(group_us) / syrk_launches           # average launch duration
    else:
        syrk_groups = None
        syrk_launches = 1
        syrk_us_alone = dev.time_kernel(_lib.SGP_T_SYRK, 10, stream)
    syrk_full_us = (dev.time_kernel(_lib.SGP_T_SYRK, 10, stream) if plan else syrk_us_alone) if not skip_alone else float("nan")
    gram_us_alone = dev.time_kernel(_lib.SGP_T_GRAM, 10, stream) if not skip_alone else float("nan")
    syrk_us = tick_us(_lib.SGP_T_SYRK)
    achieved = syrk_flops / syrk_launches / (syrk_us_alone * 1e-6) / 1e12
    traffic, traffic_src = pmc_traffic(args.workload, world)
    import ctypes as C
    sclk = C.c_double()
    _lib.check(dev._lib.sgp_measure_sclk_mhz(local_rank, C.byref(sclk)), None, "sgp_measure_sclk_mhz")
    clocks = (C.c_double * 4)()
    _lib.check(dev._lib.sgp_measure_clocks(local_rank, clocks), None, "sgp_measure_clocks")
    sclk_mfma, mfma_probe_tflops, n_cus = clocks[0], clocks[1], int(clocks[3])
    peak_at_clock = n_cus * 4 * 32.0 * sclk_mfma * 1e6 / 1e12     # 4 SIMDs x (2048 flop / 64 cycles) per CU at the measured clock
    f1_us = tick_us(_lib.SGP_T_FINISH1)
    Qp = (M + 63) // 64 * 64
    pack_doubles = (Qp // 64) * (Qp // 64 + 1) // 2 * 4096 + Qp + 8 + 1      # [lower 64 x 64 tiles | B | scalars | Ryy]
    chain_floor_us = Qp * PIVOT_CYCLES / sclk.value              # the pivots of one factorisation, nothing else

    out = {
        "metric": "VMP iterations/sec (sparse-GP node sweep, kin40k-shaped synthetic)",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "blocks": {"count": len(block_s), "steps_per_block": args.steps, "reported": "median block",
                   "ms_per_step_min": 1e3 * min(block_s) / args.steps, "ms_per_step_median": 1e3 * elapsed / args.steps,
                   "ms_per_step_max": 1e3 * max(block_s) / args.steps,
                   "iterations_per_s_over_all_blocks": len(block_s) * args.steps / sum(block_s)},
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: N={N} M={M} D={D} UniSGP regression, ARD-SE kernel at the trained kin40k "
                               f"hyper-parameters, w=1e4, prior N(0,50I), jitter 0",
                   "points_per_gpu": n_loc,
                   "parallelism": f"data-sharded x{world}, " + (f"{len(plan)} all-reduces per sweep (one per statistics group) of the packed "
                                                                f"exchange buffer, {pack_doubles} f64 in all" if (plan and sweep.hooked) else
                                                                f"1 all-reduce of the packed exchange buffer ({pack_doubles} f64)"),
                   "variant": "W' trace form (SURVEY.md Appendix A): sum I1 / sum I2 from the reduced statistics, no per-point "
                              "TRSM / TRMM (their 2 n M^2 flop are not part of the timed sweep)",
                   "collective": {"backend": sweep.backend, "world_size_seen": sweep.world,
                                  "where": "inside sgp_sweep (C ABI all-reduce hook) on the sweep's stream" if sweep.hooked
                                           else "none (single rank)"}},
        "host_binding": host_binding,
        "sclk_mhz": sclk.value,
        "sclk_mhz_under_mfma_f64": sclk_mfma,
        "roofline": {"kernel": "k_syrk_direct (Psi2 = K_uf K_uf^T, v_mfma_f64_16x16x4_f64, operands straight from global memory; k_syrk_stream where the SYRK does not fill the chip)", "bound": "mfma",
                     "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "clock": "HIP events around 10 eager launches of each of the sweep's SYRK launches alone, on the stream and the "
                              "CUs it uses inside the sweep (the duration rocprofv3 --kernel-trace --stats reports for it)",
                     "launches_per_sweep": syrk_launches, "launch_us": syrk_us_alone,
                     "first_in_to_last_out_us_in_timed_region": syrk_us, "sweeps_averaged": int(n_counted),
                     "algorithmic_flops_per_launch": syrk_flops / syrk_launches, "algorithmic_flops_per_sweep": syrk_flops,
                     "groups": syrk_groups,
                     "single_launch_all_tiles_all_cus": {"launch_us": syrk_full_us, "tflops": syrk_flops / (syrk_full_us * 1e-6) / 1e12,
                                                         "frac": syrk_flops / (syrk_full_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                                         "note": "the same kernel as ONE launch (the plain order of the sweep: multi-GPU, "
                                                                 "caller's stream, very large N); not what the timed sweeps ran when `groups` is set"},
                     "peak_at_measured_clock": peak_at_clock, "frac_of_peak_at_measured_clock": achieved / peak_at_clock if peak_at_clock > 0 else None,
                     "mfma_probe_tflops": mfma_probe_tflops,
                     "frac_of_mfma_probe": achieved / mfma_probe_tflops if mfma_probe_tflops > 0 else None,
                     "peak_note": "78.6 = MI355X FP64 matrix spec (256 CUs x 4 SIMDs x 2048 flop / 64 cycles at 2.4 GHz); "
                                  "peak_at_measured_clock = the same arithmetic at the shader clock the chip holds under "
                                  "back-to-back v_mfma_f64 (sclk_mhz_under_mfma_f64, sgp_measure_clocks); mfma_probe_tflops = "
                                  "what that back-to-back loop itself attains on this box (accumulators pinned in registers: 0.98 - "
                                  "0.99 of spec; the 45 - 48 TFLOP/s quoted through round 3 measured a loop the compiler had filled "
                                  "with accumulator copies, profiles/r04_ab_log.txt [14])"},
        # the sweep's critical path is the Lambda factorisation chain: latency-bound, priced against its pivot floor
        "roofline_chain": {"kernel": "k_potrf_step x (M/64 + 1) + k_trmv_mu_scan (Lambda = L L^T, inverse factor, Sigma rows, t, mu)",
                           "bound": "latency (dependent pivot chain)", "achieved_us": f1_us, "floor_us": chain_floor_us,
                           "overlapped_with_statistics": bool(plan),
                           "frac": chain_floor_us / f1_us if f1_us > 0 else None,
                           "model": f"{Qp} pivots x {PIVOT_CYCLES} dependent cycles at sclk"},
        # second data-sized kernel (SURVEY.md §8d asks for HBM GB/s on K_uf): k_gram_uf writes 8 n Mp bytes of K_uf once
        "roofline_k_uf": {"kernel": "k_gram_uf (K_uf assembly, coalesced 32-B stores)", "bound": "hbm",
                          "achieved": 8.0 * n_loc * dev.stats_layout()[2] / (tick_us(_lib.SGP_T_GRAM) * 1e-6) / 1e9,
                          "peak": 8000.0, "unit": "GB/s",
                          "frac": 8.0 * n_loc * dev.stats_layout()[2] / (tick_us(_lib.SGP_T_GRAM) * 1e-6) / 1e9 / 8000.0,
                          "algorithmic_bytes_per_launch": 8.0 * n_loc * dev.stats_layout()[2],
                          "note": "41 MB at T; PMC WRITE_SIZE 41.6 MB.  Not store-bound: the same store pattern alone sustains 6.7 TB/s "
                                  "(tools/store_bw_probe.hip); the kernel's phases add up, profiles/r04_ab_log.txt [19]"},
        "sweep_order": ({"kind": "overlapped", "groups": plan,
                         "note": "statistics in tile-row groups (first on all CUs, the others on a CU-masked queue) while the Lambda "
                                 "chain factors the tile columns it has; `local` and `finish1_lambda_chain` overlap in time"}
                        if plan else {"kind": "plain", "note": "statistics, then the Lambda chain"}),
        "phases_us": {"sweep_device": tick_us(_lib.SGP_T_SWEEP), "gram_uf": tick_us(_lib.SGP_T_GRAM),
                      "gram_uf_alone_hip_events": gram_us_alone, "syrk": syrk_us,
                      "local": tick_us(_lib.SGP_T_LOCAL), "gap_local_to_finish": tick_us(_lib.SGP_T_GAP_LOCAL_FINISH),
                      "finish1_lambda_chain": tick_us(_lib.SGP_T_FINISH1), "finish2_traces": tick_us(_lib.SGP_T_FINISH2)},
    }

    # N > 1: the scaling regime of the design, measured in the same run (all ranks take part; after the headline's timed region)
    if args.workload == "T" and (world > 1 or os.environ.get("SGP_BENCH_SCALING") is not None):
        try:
            out.setdefault("extra", {})["scaling_workload"] = scaling_leg(world, rank, local_rank, barrier)
        except Exception as e:                                       # pragma: no cover
            if world > 1:
                raise                                                # (a rank that drops out of a collective would hang the others)
            out.setdefault("extra", {})["scaling_workload"] = {"error": repr(e)}

    if rank == 0:
        # ---- parity of what was timed + CPU baseline (rank 0; the oracle is the checker, never the product) ----
        from oracle import sgp_oracle as O
        mu, Sig, Uv = dev.posterior()
        pred = dev.predict(Xt)
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:                                      # pragma: no cover
            threadpool_limits = None
        cores = min(16, os.cpu_count() or 1)
        # Never RAISE the BLAS thread count above what it was initialised with: torch.distributed.run exports
        # OMP_NUM_THREADS=1, and lifting OpenBLAS from 1 to 16 threads afterwards segfaults inside scipy's Cholesky
        # (seen in the two-rank rehearsal).  N > 1 runs carry no CPU baseline anyway.
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            if os.environ.get(var, "").isdigit():
                cores = min(cores, int(os.environ[var]))
        if world > 1:
            threadpool_limits = None
        Lam0, xi0 = np.eye(M) / PRIOR_VAR, np.zeros(M)

        def cpu_sweep():
            return O.vmp_sweep(Xu, X, y, None, SIGMA2, ELL, W_BAR, jitter=0.0, Lambda0=Lam0, xi0=xi0)

        ctx = threadpool_limits(limits=cores) if threadpool_limits else None
        if ctx:
            ctx.__enter__()
        t_ref0 = time.perf_counter()
        ref = cpu_sweep()
        t_ref = time.perf_counter() - t_ref0
        out["parity"] = {
            "mu_v_rel_frobenius": float(np.linalg.norm(mu - ref.mu_v) / np.linalg.norm(ref.mu_v)),
            "Sigma_v_rel_frobenius": float(np.linalg.norm(Sig - ref.Sigma_v) / np.linalg.norm(ref.Sigma_v)),
            "Uv_rel_frobenius": float(np.linalg.norm(Uv - ref.Uv) / np.linalg.norm(ref.Uv)),
            "smse_test": float(O.SMSE(yt, pred)),
            "smse_test_oracle": float(O.SMSE(yt, O.predict_mean(Xu, Xt, ref.mu_v, SIGMA2, ELL))),
            "tolerance": 1e-5,
        }
        if world == 1 and not args.no_cpu_baseline:
            reps, t_cpu = (1, t_ref) if t_ref > 5.0 else (0, 0.0)   # (one sweep at N = 1e6 is the whole bounded sample)
            while t_cpu < 10.0 and reps < 100:                   # bounded sample: ~10 s of CPU work
                t1 = time.perf_counter()
                cpu_sweep()
                t_cpu += time.perf_counter() - t1
                reps += 1
            out["cpu_baseline"] = {"value": reps / t_cpu, "unit": "iterations/s", "cores": cores, "kind": "port",
                                   "sample": f"{reps} full sweeps of the same workload by oracle/sgp_oracle.py "
                                             f"(NumPy/OpenBLAS FP64, batched BLAS-3 restatement, {cores} threads)"}
            # the reference's algorithmic shape (one rank-1 M x M message + fold + TRSV per point, one thread):
            # oracle/sgp_oracle.c on two bounded samples, extrapolated linearly in N to the full workload
            try:
                if N > 100000:
                    raise RuntimeError("skipped at this size")
                from oracle import c_oracle
                ts = []
                for ns in (100, 250):
                    t1 = time.perf_counter()
                    c_oracle.vmp_sweep_perpoint(Xu, X[:ns], y[:ns], None, SIGMA2, ELL, 0.0, W_BAR, math.log(W_BAR),
                                                np.zeros(M), PRIOR_VAR * np.eye(M))
                    ts.append(time.perf_counter() - t1)
                slope = (ts[1] - ts[0]) / 150.0
                full = ts[0] + slope * (N - 100)
                out["cpu_baseline_per_point"] = {"value": 1.0 / full, "unit": "iterations/s", "cores": 1, "kind": "port",
                                                 "sample": f"oracle/sgp_oracle.c (per-point rank-1 fold as in GPnode/UniSGPnode.jl:62-73,"
                                                           f"144-158,196-216) on 100 and 250 points, M={M}; extrapolated to N={N}: "
                                                           f"{1e3 * slope:.3f} ms/point"}
            except Exception as e:                                   # pragma: no cover
                out["cpu_baseline_per_point"] = {"error": repr(e)}
            # SMSE on the real kin40k data (tests/golden): one sweep with the reference's Xu / theta_opt, first M inducing points
            try:
                if N > 100000:
                    raise RuntimeError("skipped at this size")
                gold = os.path.join(ROOT, "tests", "golden")
                fx, kd = np.load(os.path.join(gold, "kin40k_fixture.npz")), np.load(os.path.join(gold, "kin40k_data.npz"))
                s2k, ellk = O.kernel_from_theta(fx["theta_opt"], softplus_params=True)
                Mk = min(M, fx["Xu"].shape[0])
                from gaussianprocessnode_amd import SGPDevice
                with SGPDevice(len(kd["ytrain"]), Mk, D) as dk:
                    dk.set_inducing(fx["Xu"][:Mk])
                    dk.set_data(kd["xtrain"], kd["ytrain"])
                    dk.set_kernel(s2k, ellk, 0.0)
                    dk.set_prior_isotropic(PRIOR_VAR)
                    dk.set_noise([[W_BAR]])
                    dk.sweep()
                    pk = dk.predict(kd["xtest"])
                out["parity"]["smse_kin40k_real"] = {"value": float(O.SMSE(kd["ytest"], pk)), "M": int(Mk),
                                                     "note": "one full-batch sweep on the real kin40k training set at the "
                                                             "reference's theta_opt; the reference reports 0.0834 with M=600"}
            except Exception as e:                                   # pragma: no cover
                out["parity"]["smse_kin40k_real"] = {"error": repr(e)}
            if N <= 100000:
                try:
                    out.setdefault("extra", {}).update(extras(args.workload))
                except Exception as e:                               # pragma: no cover
                    out.setdefault("extra", {})["error"] = repr(e)
        if ctx:
            ctx.__exit__(None, None, None)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
