#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel calls, average and share."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f}us  per-sweep {float(r['TotalDurationNs'])/1e3/div:9.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print(f"total per sweep {tot/1e3/div:.1f} us")
