#!/bin/bash
set -e
O=gpurun_out/r4be; mkdir -p $O
D=gaussianprocessnode_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
bash tools/ab_multi.sh 3 "prev|prev|" "mirror|new|" > $O/ab_headline.txt 2>&1; cat $O/ab_headline.txt
for i in 1 2; do for v in prev new; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo "== $v"; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done; done > $O/ab_wstats.txt 2>&1
cp $D/libsgp_hip_new.so $D/libsgp_hip.so
cat $O/ab_wstats.txt
