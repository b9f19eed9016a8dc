#!/bin/bash
# round 4, GPU call M: forward-solve role on eight waves
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 4 "fin|fin|" "cur10_tvec8|cur10|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur10.txt 2>&1
grep -E "Lambda step [78] |K_uu step [8]|gemm32|trmv|scalars|step 8:" $O/sweep_trace_cur10.txt
echo done
