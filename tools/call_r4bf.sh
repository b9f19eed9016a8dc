#!/bin/bash
set -e
O=gpurun_out/r4bf; mkdir -p $O
D=gaussianprocessnode_amd/csrc
for i in 1 2 3; do for v in 0 auto; do echo "== SGP_INTERLEAVE=$v"; if [ $v = 0 ]; then export SGP_INTERLEAVE=0; else unset SGP_INTERLEAVE; fi; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done; done > $O/ab_wstats.txt 2>&1
cat $O/ab_wstats.txt
unset SGP_INTERLEAVE
STEPS=20 bash tools/ab_multi.sh 4 "never|new|SGP_INTERLEAVE=0" "auto|new|SGP_X=1" > $O/ab_steps20.txt 2>&1; cat $O/ab_steps20.txt
bash tools/ab_multi.sh 2 "never|new|SGP_INTERLEAVE=0" "auto|new|SGP_X=1" > $O/ab_steps1000.txt 2>&1; cat $O/ab_steps1000.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
