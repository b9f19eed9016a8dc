#!/bin/bash
# round 4, GPU call AJ: forced overlap plans at C2 (M = 256: 4 tile columns) and at 3000 x 512, 10000 x 128 now that they are gated
O=gpurun_out/r4aj; mkdir -p $O
EXTRA_ARGS="--workload C2" STEPS=500 bash tools/ab_multi.sh 2 "C2-auto|g10k|" "C2-cut1|g10k|SGP_OVERLAP_COLS=1" "C2-cut2|g10k|SGP_OVERLAP_COLS=2" "C2-cut1,2|g10k|SGP_OVERLAP_COLS=1,2" "C2-cut1,3|g10k|SGP_OVERLAP_COLS=1,3" 2>&1 | tee $O/ab_C2.txt
EXTRA_ARGS="--workload N5K" STEPS=500 bash tools/ab_multi.sh 2 "N5K-auto|g10k|" "N5K-cut2,5|g10k|SGP_OVERLAP_COLS=2,5" "N5K-cut1|g10k|SGP_OVERLAP_COLS=1" "N5K-cut3|g10k|SGP_OVERLAP_COLS=3" "N5K-cut1,4|g10k|SGP_OVERLAP_COLS=1,4" 2>&1 | tee $O/ab_N5K.txt
