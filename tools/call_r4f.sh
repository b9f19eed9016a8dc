#!/bin/bash
# round 4, GPU call F: wide SYRK + early B + statistics enqueued first; variants
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "narrow_r4b|new|" "wide_only|w16g|" "cur|cur|" "cur_no_early_b|cur|SGP_EARLY_B=0" "cur_wide_g0_only|cur|SGP_SYRK_WIDE=2" "cur_cut4|cur|SGP_OVERLAP_COLS=4" "cur_cut2|cur|SGP_OVERLAP_COLS=2" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur.txt 2>&1
head -45 $O/sweep_trace_cur.txt | grep -E "syrk|assemble|gram|Lambda step [0-3] |join_wait|has its"
cp gaussianprocessnode_amd/csrc/libsgp_hip_new.so /tmp/keep_new.so
timeout -k 10 200 python tools/wstats_time.py 2>&1 | tee $O/wstats_time_cur.txt
cp gaussianprocessnode_amd/csrc/libsgp_hip.so /tmp/keep_cur.so; cp /tmp/keep_new.so gaussianprocessnode_amd/csrc/libsgp_hip.so
timeout -k 10 200 python tools/wstats_time.py 2>&1 | tee $O/wstats_time_narrow_r4b.txt
cp /tmp/keep_cur.so gaussianprocessnode_amd/csrc/libsgp_hip.so
echo done
