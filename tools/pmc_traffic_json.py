#!/usr/bin/env python3
"""profiles/r04_pmc_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command (run with
SGP_BENCH_SKIP_ALONE=1: every k_syrk_stream launch is then one of the timed sweeps' own): the HBM / fabric bytes per k_syrk_stream
launch that bench.py reports as roofline.traffic -- the average over the sweep's launches (one per statistics group of the
overlapped sweep), with the per-grid breakdown.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane
streaming reads (gfx950 tallies 128-B requests at 64 B)."""
import csv, json, subprocess, sys, collections


by_grid = {}


def avg_kib(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_syrk_" in r["Kernel_Name"]:
            acc[0].append(float(r["Counter_Value"]))
            acc[("grid", int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 256) or 256), 1))].append(float(r["Counter_Value"]))
    v = acc[0]
    by_grid[counter] = {str(k[1]): {"launches": len(x), "avg_kib": sum(x) / len(x)} for k, x in acc.items() if k != 0}
    return sum(v) / len(v), len(v)


fetch_csv, write_csv, workload, out = sys.argv[1:5]
f, nf = avg_kib(fetch_csv, "FETCH_SIZE")
w, nw = avg_kib(write_csv, "WRITE_SIZE")
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
except Exception:
    commit = "unknown"
rec = {"workload": workload, "n_gpus": 1, "kernel": "k_syrk_direct (k_syrk_stream where the SYRK does not fill the chip)",
       "fetch_size_bytes": f * 1024, "write_size_bytes": w * 1024, "fetch_correction": 2.0,
       "traffic_bytes_per_launch": 2.0 * f * 1024 + w * 1024, "launches_averaged": [nf, nw], "by_grid_workgroups": by_grid, "commit": commit,
       "source": "SGP_BENCH_SKIP_ALONE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
