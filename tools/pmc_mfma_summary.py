#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 counter_collection.csv of
   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE  (profiles/r01_pmc_mfma_T.txt)."""
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline")
print("# per-launch averages; counters are summed over the 1024 SIMDs (SQ) / 8 XCDs (GRBM).  v_mfma_f64_16x16x4_f64 = 4 MOPS, busy 64 cycles.")
print("# MfmaUtil = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8);  CU-busy share = (BUSY_CU / 256) / (GUI_ACTIVE / 8)   [kernels run one at a time under --pmc]")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    av = {m: sum(v) / len(v) for m, v in c.items()}
    if av["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
        continue
    gui = av["GRBM_GUI_ACTIVE"] / 8
    print(f"{k[:28]:28s} calls {n:4d}  MFMA_BUSY {av['SQ_VALU_MFMA_BUSY_CYCLES']:12.0f}  MOPS_F64 {av['SQ_INSTS_VALU_MFMA_MOPS_F64']:10.0f}  "
          f"BUSY_CU {av['SQ_BUSY_CU_CYCLES']:11.0f}  GUI_ACTIVE/XCD {gui:9.0f}  MfmaUtil {av['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:.3f}  "
          f"CU-busy {av['SQ_BUSY_CU_CYCLES'] / 256 / gui:.3f}")
