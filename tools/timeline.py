#!/usr/bin/env python3
"""Print the kernel timeline of the last graph replay in a rocprofv3 kernel_trace.csv (start/end relative, queue)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# sweeps are delimited by k_scalars (the last kernel of a sweep)
idx = [i for i, r in enumerate(rows) if "k_scalars" in r["Kernel_Name"]]
start = idx[-3] + 1 if len(idx) > 2 else 0
end = idx[-2] + 1 if len(idx) > 2 else len(rows)
t0 = int(rows[start]["Start_Timestamp"])
if start > 0:
    print(f"(previous sweep's k_scalars ended {(t0 - int(rows[start - 1]['End_Timestamp'])) / 1e3:.1f} us before this sweep's first kernel)")
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}us q{r['Queue_Id']:>3s} {r['Kernel_Name'][:40]}")
