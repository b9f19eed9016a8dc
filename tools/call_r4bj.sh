#!/bin/bash
set -e
O=gpurun_out/r4bj; mkdir -p $O
STEPS=20 bash tools/ab_multi.sh 8 "never|new|SGP_INTERLEAVE=0" "streak|new|SGP_X=1" > $O/ab_steps20.txt 2>&1; cat $O/ab_steps20.txt
for i in 1 2 3; do for v in 0 auto; do echo "== SGP_INTERLEAVE=$v"; if [ $v = 0 ]; then export SGP_INTERLEAVE=0; else unset SGP_INTERLEAVE; fi; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done; done > $O/ab_wstats.txt 2>&1
cat $O/ab_wstats.txt
