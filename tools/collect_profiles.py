#!/usr/bin/env python3
"""Copy what tools/measure_round.sh left under gpurun_out/final4/ into profiles/r04_* (the tracked, judged copies) and stamp
the traffic record with the commit (the GPU box has no .git).  Prints the headline figures."""
import json, os, re, shutil, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, P = os.path.join(R, "gpurun_out", "final4"), os.path.join(R, "profiles")
names = {"bench_T.json": "r04_bench_T.json", "bench_N1M.json": "r04_bench_N1M.json", "kt.json": "r04_bench_T_under_rocprof.json",
         "kernel_stats.csv": "r04_bench_T_kernel_stats.csv", "kernel_stats_per_sweep.txt": "r04_bench_T_kernel_stats_per_sweep.txt",
         "sweep_timeline_T.txt": "r04_sweep_timeline_T.txt", "sweep_timeline_T_plain_order.txt": "r04_sweep_timeline_T_plain_order.txt",
         "sweep_timeline_C3.txt": "r04_sweep_timeline_C3.txt", "step_trace_T.txt": "r04_step_trace_T.txt", "syrk_launches.txt": "r04_syrk_launches.txt",
         "bench_T_plain_order.json": "r04_bench_T_plain_order.json", "pmc_valu.txt": "r04_pmc_valu_T.txt",
         "hooked_train.json": "r04_hooked_train.json", "hooked_train_kernel_stats.txt": "r04_hooked_train_kernel_stats.txt",
         "soak.txt": "r04_soak.txt", "rehearse_two_ranks.txt": "r04_rehearse_two_ranks.txt",
         "pmc_FETCH_SIZE.txt": "r04_pmc_fetch_size_T.txt", "pmc_WRITE_SIZE.txt": "r04_pmc_write_size_T.txt",
         "pmc_mfma.txt": "r04_pmc_mfma_T.txt", "config_rates.txt": "r04_config_rates.txt", "accuracy_sweep.txt": "r04_accuracy_sweep.txt",
         "train_kin40k.txt": "r04_train_kin40k.json", "train_banana.txt": "r04_train_banana.json", "pytest_gpu.txt": "r04_pytest_gpu.txt",
         "rehearse_two_ranks_overlapped.txt": "r04_rehearse_two_ranks_overlapped.txt", "wstats_time.txt": "r04_wstats_time.txt",
         "hooked_train_rc.txt": "r04_hooked_train_rc.txt", "ab_r3_vs_r4.txt": "r04_ab_r3_vs_r4_measurement_box.txt",
         "bench_T_syrk_256_threads.json": "r04_bench_T_syrk_256_threads.json", "bench_driver_style.json": "r04_bench_T_driver_style_steps20.json",
         "mfma_f64_probe.txt": "r04_mfma_f64_probe.txt", "dpp_f64_probe.txt": "r04_dpp_f64_probe.txt",
         "syrk_direct_probe.txt": "r04_syrk_direct_probe.txt", "store_bw_probe.txt": "r04_store_bw_probe.txt",
         }
for a, b in names.items():
    src = os.path.join(F, a)
    if not os.path.exists(src):
        print("missing", a); continue
    txt = "".join(l for l in open(src) if "amdgpu.ids" not in l)
    open(os.path.join(P, b), "w").write(txt)
c = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, cwd=R).strip()
d = json.load(open(os.path.join(F, "pmc_traffic.json")))
d["commit"] = c + " (library and bench as of this commit; the GPU box has no .git, stamped when the summary was copied)"
json.dump(d, open(os.path.join(P, "r04_pmc_traffic.json"), "w"), indent=1)
last = lambda f: json.loads(open(os.path.join(P, f)).read().strip().splitlines()[-1])
b, n, t = last("r04_bench_T.json"), last("r04_bench_N1M.json"), last("r04_train_kin40k.json")
r, p = b["roofline"], b["phases_us"]
print(f"T: {b['value']:.0f} it/s, wall {b['ms_per_step']*1e3:.1f} us, device {p['sweep_device']:.1f}, local {p['local']:.1f} (gram {p['gram_uf']:.1f}, "
      f"syrk {p['syrk']:.1f}), F1 {p['finish1_lambda_chain']:.1f}, F2 {p['finish2_traces']:.1f}")
print(f"   syrk launches (avg of {r['launches_per_sweep']}) {r['launch_us']:.1f} us = {r['achieved']:.1f} TF, frac {r['frac']:.3f}, of probe {r['frac_of_mfma_probe']:.2f} ({r['mfma_probe_tflops']:.1f} TF at "
      f"{b['sclk_mhz_under_mfma_f64']:.0f} MHz); chain frac {b['roofline_chain']['frac']:.3f} floor {b['roofline_chain']['floor_us']:.1f}; cpu {b['cpu_baseline']['value']:.2f}")
print(f"N1M: {n['value']:.1f} sweeps/s, gram {n['phases_us']['gram_uf']/1e3:.2f} ms, syrk {n['phases_us']['syrk']/1e3:.2f} ms, local {n['phases_us']['local']/1e3:.2f} ms")
print(f"kin40k training: {t['train_seconds']:.2f} s")
print(open(os.path.join(P, "r04_syrk_launches.txt")).read().splitlines()[2])
print(open(os.path.join(P, "r04_config_rates.txt")).read())
