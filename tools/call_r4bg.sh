#!/bin/bash
set -e
O=gpurun_out/r4bg; mkdir -p $O
for i in 1 2 3; do for v in 0 auto; do echo "== SGP_INTERLEAVE=$v"; if [ $v = 0 ]; then export SGP_INTERLEAVE=0; else unset SGP_INTERLEAVE; fi; timeout -k 10 100 python tools/host_enqueue_time.py 2>&1 | grep -v amdgpu; done; done > $O/host_enqueue.txt 2>&1
cat $O/host_enqueue.txt
