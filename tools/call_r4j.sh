#!/bin/bash
# round 4, GPU call J: the launch behind the last Cholesky step (all operands of the finishing roles requested up front)
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "cur3|cur3|" "cur5|cur5|" "cur6_step8|cur6|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur6.txt 2>&1
grep -E "syrk|assemble|gram|Lambda step [0-8] |K_uu step [08]|gemm32|trmv|scalars|step 8:" $O/sweep_trace_cur6.txt
echo done
