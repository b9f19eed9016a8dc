#!/bin/bash
# round 4, GPU call I: dual-path k_assemble; per-step log-determinant partials
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "cur3|cur3|" "cur4_dual_assemble|cur4|" "cur5_logdet_parts|cur5|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur5.txt 2>&1
head -48 $O/sweep_trace_cur5.txt | grep -E "syrk|assemble|gram|Lambda step [0-8] |K_uu step [08]|gemm32|trmv|scalars"
echo "== config rates, cur5"; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu | tee $O/config_rates_cur5.txt
timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu | tee $O/wstats_time_cur5.txt
echo done
