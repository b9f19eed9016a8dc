#!/bin/bash
set -e
O=gpurun_out/r4bi; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['blocks']; p=d['phases_us']
print('auto', round(d['value'],1), 'blocks us/sweep min/med/max', round(1e3*b['ms_per_step_min'],1), round(1e3*b['ms_per_step_median'],1), round(1e3*b['ms_per_step_max'],1), 'device', round(p['sweep_device'],1), 'local', round(p['local'],1), 'gram', round(p['gram_uf'],1), 'syrk', round(p['syrk'],1))"; done > $O/auto_blocks.txt 2>&1
cat $O/auto_blocks.txt
