#!/bin/bash
O=gpurun_out/r4bq; mkdir -p $O
for i in 1 2 3; do for v in default always; do echo "== $v"; if [ $v = always ]; then export SGP_INTERLEAVE=1; else unset SGP_INTERLEAVE; fi; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu; done; done > $O/config_rates_ab.txt 2>&1
unset SGP_INTERLEAVE
cat $O/config_rates_ab.txt
for w in C2 C3; do for i in 1 2 3; do for v in default always; do if [ $v = always ]; then export SGP_INTERLEAVE=1; else unset SGP_INTERLEAVE; fi; SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w steps20', '$v', round(d['value'],1))"; done; done; done > $O/steps20_other.txt 2>&1
cat $O/steps20_other.txt
