#!/usr/bin/env python3
"""Print the kernel timeline of the last graph replay in a rocprofv3 kernel_trace.csv (start/end relative, queue)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find last k_stamp_reset (start of a sweep)
idx = [i for i, r in enumerate(rows) if "k_stamp_reset" in r["Kernel_Name"]]
start = idx[-2] if len(idx) > 1 else idx[-1]
end = idx[-1] if len(idx) > 1 else len(rows)
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}us q{r['Queue_Id']:>3s} {r['Kernel_Name'][:40]}")
