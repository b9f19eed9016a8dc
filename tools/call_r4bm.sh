#!/bin/bash
O=gpurun_out/r4bm; mkdir -p $O
run() { SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['blocks']; p=d['phases_us']
print('$TAG', round(d['value'],1), 'blocks min/med/max', round(1e3*b['ms_per_step_min'],1), round(1e3*b['ms_per_step_median'],1), round(1e3*b['ms_per_step_max'],1), 'device', round(p['sweep_device'],1), d.get('host_binding'))"; }
for i in 1 2 3 4 5 6 7 8; do
TAG=bound run
export SGP_NO_HOST_BIND=1; TAG=free run; unset SGP_NO_HOST_BIND
done > $O/runs.txt 2>&1
cat $O/runs.txt; tail -3 $O/err.txt
