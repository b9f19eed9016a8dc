#!/bin/bash
O=gpurun_out/r4bn; mkdir -p $O
for i in 1 2 3 4; do for v in 0 auto; do echo "== SGP_INTERLEAVE=$v"; if [ $v = 0 ]; then export SGP_INTERLEAVE=0; else unset SGP_INTERLEAVE; fi; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v "amdgpu\|host binding"; done; done > $O/ab_wstats.txt 2>&1
cat $O/ab_wstats.txt
