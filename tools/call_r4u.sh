#!/bin/bash
# round 4, GPU call U: where to cut the tile columns now that the SYRK launches are shorter; the repaired MFMA probe
O=gpurun_out/r4u; mkdir -p $O
timeout -k 10 60 ./tools/mfma_f64_probe > $O/mfma_f64_probe.txt 2>&1; grep -E "acc=4|acc=8|valu" $O/mfma_f64_probe.txt
bash tools/ab_multi.sh 2 "cut3|dir2|" "cut2|dir2|SGP_OVERLAP_COLS=2" "cut4|dir2|SGP_OVERLAP_COLS=4" "cut2,5|dir2|SGP_OVERLAP_COLS=2,5" "cut3,6|dir2|SGP_OVERLAP_COLS=3,6" "cut2,4|dir2|SGP_OVERLAP_COLS=2,4" 2>&1 | tee $O/ab_cuts.txt
SGP_LIB_VARIANT=trace SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace.txt 2>&1; head -45 $O/sweep_trace.txt
