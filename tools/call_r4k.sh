#!/bin/bash
# round 4, GPU call K: Sigma's row Tn-2 in the product launch; polled host waits
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "cur3|cur3|" "cur6|cur6|" "cur8|cur8|" "cur9_b_in_chain|cur9|" "cur9_b_in_assemble|cur9|SGP_B_IN_CHAIN=0" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur9.txt 2>&1
grep -E "gram|syrk|assemble|Lambda step [0178] |K_uu step [08] |gemm32|trmv|scalars|step [08]:" $O/sweep_trace_cur9.txt
for sw in 1 0; do echo "== SGP_SPIN_WAIT=$sw"; SGP_SPIN_WAIT=$sw timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu | tee $O/wstats_time_spin$sw.txt; done
echo "== config rates, cur9"; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu | tee $O/config_rates_cur9.txt
echo done
