#!/bin/bash
# round 4, GPU call BA: runtime knobs against the slow host phase after an idle period (AMD_DIRECT_DISPATCH, HSA_ENABLE_INTERRUPT)
O=gpurun_out/r4ba; mkdir -p $O
for envs in "" "HSA_ENABLE_INTERRUPT=0" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_INTERRUPT=0 AMD_DIRECT_DISPATCH=0" ""; do
  echo "== env: $envs"
  env $envs timeout -k 10 100 python tools/host_enqueue_time.py 2>&1 | grep -v amdgpu | tail -3
  env $envs SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench --steps 20:', round(d['value'],1), 'device', round(d['phases_us']['sweep_device'],1))"
  env $envs WSTATS_SHORT=1 timeout -k 10 200 python tools/wstats_copy_cost.py 2>&1 | grep "fresh arrays" | tail -1
done | tee $O/runtime_knobs.txt
