#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc counter_collection.csv (values in KiB for FETCH_SIZE / WRITE_SIZE)."""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:44]:44s} {c:12s} calls {len(v):5d}  avg {sum(v)/len(v):12.1f} KiB  = {sum(v)/len(v)*1024/1e6:9.3f} MB")
