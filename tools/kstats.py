#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel calls, average and share.
    kstats.py kernel_stats.csv [sweeps | auto]      (auto: one k_scalars launch per sweep)
Two kinds of kernels are listed apart because their durations are WAITS, not work: k_join_wait (one wave spinning on a device word
until another stream's kernel sets it) and the K_uu chain's k_prep_xu (its first kernel, which waits for the previous sweep's done
word and for the SYRK's gate).  They overlap with the kernels they wait for, so the column total with them included is not the
time of a sweep -- the sweep's device time is bench.py's phases_us.sweep_device (in-kernel stamps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
arg = sys.argv[2] if len(sys.argv) > 2 else "1"
if arg == "auto":
    div = float(next((r['Calls'] for r in rows if r['Name'].startswith('sgp::k_scalars') or 'k_scalars' in r['Name']), 1))
else:
    div = float(arg)
is_wait = lambda name: ('k_join_wait' in name) or ('k_prep_xu' in name)
tot = sum(float(r['TotalDurationNs']) for r in rows)
work = sum(float(r['TotalDurationNs']) for r in rows if not is_wait(r['Name']))
print(f"(per-sweep columns: totals divided by {div:.0f} sweeps)")
for r in rows:
    tag = "  [wait, not work]" if is_wait(r['Name']) else ""
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f}us  per-sweep {float(r['TotalDurationNs'])/1e3/div:9.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%{tag}")
print(f"kernel time per sweep, waits excluded: {work/1e3/div:.1f} us (the kernels of three streams overlap: this is a sum of durations, not the sweep's "
      f"time); with the spinning waits: {tot/1e3/div:.1f} us")
