#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by 2-3 %): alternates csrc/libsgp_hip_<name>.so into place and runs
# the bench line.   gpurun -- 'bash tools/ab_libs.sh prev new [rounds] [bench args]'
A=$1; B=$2; R=${3:-3}; shift 3
D=gaussianprocessnode_amd/csrc
for i in $(seq $R); do
  for v in $A $B; do
    cp $D/libsgp_hip_$v.so $D/libsgp_hip.so
    SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1000 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'chain', round(d['phases_us']['finish1_lambda_chain'],1))"
  done
done
