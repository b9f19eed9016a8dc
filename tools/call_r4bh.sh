#!/bin/bash
set -e
O=gpurun_out/r4bh; mkdir -p $O
STEPS=20 bash tools/ab_multi.sh 7 "never|new|SGP_INTERLEAVE=0" "auto|new|SGP_X=1" "always|new|SGP_INTERLEAVE=1" > $O/ab_steps20.txt 2>&1; cat $O/ab_steps20.txt
