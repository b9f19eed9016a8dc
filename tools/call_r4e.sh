#!/bin/bash
# round 4, GPU call E: 16-wave SYRK with group-local LDS barriers; per-point path timing
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweep or overlapped or determinism or stats" > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "narrow|new|" "wide_wg_barrier|w16|" "wide_group_barrier|w16g|" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_wide_group_barrier.txt 2>&1
grep -E "syrk|assemble|gram_uf|Lambda step 0" $O/sweep_trace_wide_group_barrier.txt | head
timeout -k 10 200 python tools/wstats_time.py 2>&1 | tee $O/wstats_time.txt
echo done
