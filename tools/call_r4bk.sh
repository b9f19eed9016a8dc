#!/bin/bash
O=gpurun_out/r4bk; mkdir -p $O
python tools/affinity_probe.py > $O/affinity.txt 2>&1; cat $O/affinity.txt
run() { SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 "$@" bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['blocks']; p=d['phases_us']
print('$TAG', round(d['value'],1), 'blocks min/med/max', round(1e3*b['ms_per_step_min'],1), round(1e3*b['ms_per_step_median'],1), round(1e3*b['ms_per_step_max'],1), 'device', round(p['sweep_device'],1))"; }
for i in 1 2 3 4 5 6 7 8; do
TAG=default run python
TAG=show run python -c "import os,sys; print('cpu', os.sched_getcpu(), file=sys.stderr); sys.argv=sys.argv[1:]; exec(open('bench.py').read())" 
done > $O/runs.txt 2>&1
cat $O/runs.txt
