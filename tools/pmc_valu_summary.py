#!/usr/bin/env python3
"""VALU occupancy per kernel from a rocprofv3 counter_collection.csv of
   --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
(is k_gram_uf bound by its FP64 vector work, as DESIGN.md says from instruction counts?)."""
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline")
print("# per-launch averages; SQ counters are summed over the 1024 SIMDs, SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles (x 4 = cycles),")
print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs.  VALUBusy = 4 x ACTIVE_INST_VALU / 1024 / (GUI_ACTIVE / 8): the share of the launch's")
print("# cycles in which a SIMD's vector ALU is executing (FP64 transcendentals and FMAs hold it for several cycles per instruction).")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_ACTIVE_INST_VALU", [0]))):
    if "SQ_ACTIVE_INST_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    av = {m: sum(v) / len(v) for m, v in c.items()}
    gui = av["GRBM_GUI_ACTIVE"] / 8
    if gui <= 0 or av["SQ_ACTIVE_INST_VALU"] == 0:
        continue
    print(f"{k[:28]:28s} calls {len(c['SQ_ACTIVE_INST_VALU']):4d}  INSTS_VALU {av.get('SQ_INSTS_VALU', 0):12.0f}  ACTIVE_INST_VALU(quad-cyc) {av['SQ_ACTIVE_INST_VALU']:12.0f}  "
          f"WAVE_CYCLES(quad-cyc) {av.get('SQ_WAVE_CYCLES', 0):13.0f}  GUI_ACTIVE/XCD {gui:9.0f}  VALUBusy {4 * av['SQ_ACTIVE_INST_VALU'] / 1024 / gui:.3f}  "
          f"CU-busy {av.get('SQ_BUSY_CU_CYCLES', 0) / 256 / gui:.3f}")
