#!/bin/bash
# round 4, GPU call AD: counters of k_gram_uf at N = 10^6 (what is it waiting for?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4ad; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -iE "WRREQ|WR_REQ|WRITE|STALL|WAIT|BUSY|LEVEL|OCCUP" $O/avail.txt | head -150 > $O/avail_filtered.txt
wc -l $O/avail.txt $O/avail_filtered.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_WRITE_sum GRBM_GUI_ACTIVE" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum TA_ADDR_STALL_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pg$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg$i -- python3 $R/tools/gram_alone.py > $O/run$i.txt 2>&1
  h=$(find /tmp/pg$i -name "*counter_collection.csv" | head -1)
  if [ -n "$h" ]; then python3 - "$h" <<'PY' > $O/counters$i.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_gram_uf" in r["Kernel_Name"]: acc[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
for k, d in acc.items(): print(f"{k:40s} launches {len(d['v']):3d}  avg {sum(d['v'])/len(d['v']):.4g}")
PY
  cat $O/counters$i.txt; else echo "set $i: no counter file"; tail -3 $O/run$i.txt; fi
done
