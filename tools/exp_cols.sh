#!/bin/bash
# sweeps/s of one workload under several tile-column groupings of the overlapped sweep (SGP_OVERLAP_COLS), one box
#   gpurun -- 'bash tools/exp_cols.sh C3 "" 0 "4,6" "2,4,6" ...'      ("" = the planner's choice, 0 = plain order)
W=$1; shift
for cols in "$@"; do
  if [ "$cols" = "0" ]; then export SGP_OVERLAP=0; unset SGP_OVERLAP_COLS; else export SGP_OVERLAP=1; export SGP_OVERLAP_COLS=$cols; fi
  [ -z "$cols" ] && { unset SGP_OVERLAP; unset SGP_OVERLAP_COLS; }
  SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $W --steps 300 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cols=[$cols]', round(d['value'],1), 'sweep_us', round(d['phases_us']['sweep_device'],1))"
done
