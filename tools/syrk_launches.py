#!/usr/bin/env python3
"""Durations of every k_syrk_stream launch in a rocprofv3 kernel trace (csv), in launch order: the launches inside the timed
sweeps first, then the eager launches of the kernel alone that bench.py times with HIP events (its `roofline.launch_us`).
Usage: syrk_launches.py <kernel_trace.csv> [n_alone=10]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_syrk_stream" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n_alone = int(sys.argv[2]) if len(sys.argv) > 2 else 10
alone, sweeps = dur[-n_alone:], dur[:-n_alone]
mean = lambda v: sum(v) / max(len(v), 1)
print(f"k_syrk_stream: {len(dur)} launches, average {mean(dur):.2f} us (what --stats reports)")
print(f"  in the sweeps (first {len(sweeps)}): average {mean(sweeps):.2f} us, min {min(sweeps):.2f}, max {max(sweeps):.2f}")
print(f"  alone, after the timed region (last {len(alone)}; bench.py's HIP-event launches): average {mean(alone):.2f} us, "
      f"min {min(alone):.2f}, max {max(alone):.2f}")
print("all launches (us):", " ".join(f"{d:.1f}" for d in dur))
