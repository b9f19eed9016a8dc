#!/bin/bash
O=gpurun_out/r4bv; mkdir -p $O
bash tools/ab_multi.sh 2 "default_2_5|new|SGP_X=1" "c2_4|new|SGP_OVERLAP_COLS=2,4" "c2_6|new|SGP_OVERLAP_COLS=2,6" "c3_5|new|SGP_OVERLAP_COLS=3,5" "c3_6|new|SGP_OVERLAP_COLS=3,6" "c1_4|new|SGP_OVERLAP_COLS=1,4" "c2|new|SGP_OVERLAP_COLS=2" "c3|new|SGP_OVERLAP_COLS=3" "c1_3_5|new|SGP_OVERLAP_COLS=1,3,5" "c2_4_6|new|SGP_OVERLAP_COLS=2,4,6" > $O/cuts.txt 2>&1; cat $O/cuts.txt
