#!/bin/bash
O=gpurun_out/r4bp; mkdir -p $O
STEPS=20 bash tools/ab_multi.sh 6 "streak3|new|SGP_X=1" "first_of_block|new|SGP_INTERLEAVE_STREAK=1" "always|new|SGP_INTERLEAVE=1" > $O/ab_steps20.txt 2>&1; cat $O/ab_steps20.txt
