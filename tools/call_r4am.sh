#!/bin/bash
# round 4, GPU call AM: LDS counters of every kernel of the sweep at T (bank conflicts?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4am; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl
SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pl -- python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline > /dev/null 2> $O/err.txt
h=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
python3 - "$h" <<'PY' | tee $O/lds_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':42s} {'launches':>8s} {'GUI/XCD':>10s} {'LDS_IDX_ACTIVE':>14s} {'BANK_CONFLICT':>14s} {'conflict share':>14s} {'INSTS_LDS':>10s} {'LDS active / CU / launch cycles':>14s}")
for k, d in sorted(acc.items()):
    n = len(d["GRBM_GUI_ACTIVE"]); avg = lambda c: sum(d[c]) / max(1, len(d[c]))
    gui = avg("GRBM_GUI_ACTIVE") / 8
    act, bc = avg("SQ_LDS_IDX_ACTIVE"), avg("SQ_LDS_BANK_CONFLICT")
    print(f"{k:42s} {n:8d} {gui:10.0f} {act:14.4g} {bc:14.4g} {bc / act if act else 0:14.2f} {avg('SQ_INSTS_LDS'):10.4g} {act / 256 / gui if gui else 0:14.3f}")
PY
