#!/usr/bin/env python3
"""Durations of every k_syrk_stream launch in a rocprofv3 kernel trace (csv), grouped by grid size: the sweep's SYRK launches (one
per statistics group of the overlapped sweep, or the single launch of the plain order) inside the timed sweeps, and the eager
launches of each alone that bench.py times with HIP events (its `roofline.launch_us` = the average over the sweep's launches).
Usage: syrk_launches.py <kernel_trace.csv>"""
import collections, csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_syrk_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.OrderedDict()
for r in rows:
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", 256) or 256), 1)
    by.setdefault(grid, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = lambda v: sum(v) / max(len(v), 1)
alld = [d for v in by.values() for d in v]
print(f"k_syrk_direct / k_syrk_stream: {len(alld)} launches, average {mean(alld):.2f} us (what --stats reports)")
for grid, d in by.items():
    print(f"  grid {grid:5d} workgroups: {len(d):4d} launches, average {mean(d):.2f} us, min {min(d):.2f}, max {max(d):.2f}; last 10 (alone, "
          f"HIP-event timed): {mean(d[-10:]):.2f} us")
