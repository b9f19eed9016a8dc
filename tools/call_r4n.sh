#!/bin/bash
# round 4, GPU call N: sgp_wait in the bench brackets (driver-style call: --steps 20), forward solve padded batch, per-point probe
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style --steps 20:', round(d['value'],1), d['blocks'])"; done | tee $O/bench_steps20.txt
bash tools/ab_multi.sh 3 "fin|fin|" "cur11|cur11|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur11.txt 2>&1
grep -E "Lambda step [78] |gemm32|trmv|scalars|step 8:" $O/sweep_trace_cur11.txt
for m in a b c d; do timeout -k 10 120 python tools/wstats_probe.py $m 2>&1 | grep -v amdgpu; done | tee $O/wstats_probe.txt
echo done
