#!/bin/bash
# round 4, GPU call H: wide SYRK + 4-rows-per-thread k_assemble + statistics enqueued first (K_uu arrangement of round 3)
O=gpurun_out/r4h; mkdir -p $O
D=gaussianprocessnode_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "narrow_r4b|new|" "wide_only|w16g|" "cur3|cur3|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur3.txt 2>&1
head -48 $O/sweep_trace_cur3.txt | grep -E "syrk|assemble|gram|Lambda step [0-8] |join_wait|K_uu step [08]|prep_xu|gemm32|trmv|scalars"
echo "== config rates, cur3"; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu | tee $O/config_rates_cur3.txt
timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu | tee $O/wstats_time_cur3.txt
echo done
