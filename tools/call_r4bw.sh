#!/bin/bash
O=gpurun_out/r4bw; mkdir -p $O
bash tools/ab_multi.sh 2 "default|new|SGP_X=1" "g1_after_0|new|SGP_G1_AFTER=0" "g1_after_1|new|SGP_G1_AFTER=1" "reserved_1|new|SGP_RESERVED_PER_SE=1" "reserved_3|new|SGP_RESERVED_PER_SE=3" "reserved_4|new|SGP_RESERVED_PER_SE=4" "wt_masked|new|SGP_SYRK_WT=1" "wt_all|new|SGP_SYRK_WT=2" "join_event|new|SGP_JOIN_EVENT=1" > $O/knobs.txt 2>&1; cat $O/knobs.txt
