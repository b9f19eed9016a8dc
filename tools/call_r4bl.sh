#!/bin/bash
O=gpurun_out/r4bl; mkdir -p $O
run() { SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 "$@" bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['blocks']; p=d['phases_us']
print('$TAG', round(d['value'],1), 'blocks min/med/max', round(1e3*b['ms_per_step_min'],1), round(1e3*b['ms_per_step_median'],1), round(1e3*b['ms_per_step_max'],1), 'device', round(p['sweep_device'],1))"; }
for i in 1 2 3 4 5 6; do
TAG=node0 run taskset -c 0-63,128-191 python
TAG=node1 run taskset -c 64-127,192-255 python
TAG=free run python
done > $O/runs.txt 2>&1
cat $O/runs.txt
TAG=node0_1000 ; SGP_BENCH_SKIP_ALONE=1 taskset -c 0-63,128-191 python bench.py --no-cpu-baseline --steps 1000 2>/dev/null | tail -c 200
echo
SGP_BENCH_SKIP_ALONE=1 taskset -c 64-127,192-255 python bench.py --no-cpu-baseline --steps 1000 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('node1 steps 1000', d['value'])"
