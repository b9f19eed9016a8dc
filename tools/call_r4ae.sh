#!/bin/bash
# round 4, GPU call AE: counters of k_quadform_fused at T (what is the per-point kernel waiting for?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4ae; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pq$i -- python3 $R/tools/quadform_alone.py > $O/run$i.txt 2>&1
  h=$(find /tmp/pq$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$h" ]; then python3 - "$h" <<'PY' > $O/counters$i.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_quadform" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(f"{k:40s} launches {len(v):3d}  avg {sum(v)/len(v):.4g}")
PY
  cat $O/counters$i.txt; else echo "set $i: no counter file"; tail -2 $O/run$i.txt | cut -c1-200; fi
done
