#!/bin/bash
# A/B of several (library build, environment) variants on ONE box, interleaved (boxes differ by 2-3 %):
#   gpurun -- 'bash tools/ab_multi.sh ROUNDS "name|libsuffix|ENV=1 ENV2=x" ...'      (libsuffix: csrc/libsgp_hip_<suffix>.so)
# Every line: name, sweeps/s of the bench's timed region, device sweep, Lambda chain (in-kernel stamps).  EXTRA_ARGS go to bench.py.
R=$1; shift
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip.so /tmp/ab_keep.so
for i in $(seq $R); do
  for spec in "$@"; do
    IFS='|' read -r name lib envs <<< "$spec"
    cp $D/libsgp_hip_$lib.so $D/libsgp_hip.so
    env $envs SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps ${STEPS:-1000} $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_us']
print('$name', round(d['value'],1), 'device', round(p['sweep_device'],1), 'chain', round(p['finish1_lambda_chain'],1), 'F2', round(p['finish2_traces'],1))"
  done
done
cp /tmp/ab_keep.so $D/libsgp_hip.so
