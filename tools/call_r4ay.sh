#!/bin/bash
# round 4, GPU call AY: interleaved enqueue on / off (SGP_NO_INTERLEAVE), many alternations, short blocks and the per-point loop
O=gpurun_out/r4ay; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip.so $D/libsgp_hip_cur.so
STEPS=20 bash tools/ab_multi.sh 6 "interleave|cur|" "old-order|cur|SGP_NO_INTERLEAVE=1" 2>&1 | tee $O/ab_steps20.txt
for i in 1 2 3; do for m in 0 1; do echo -n "no_interleave=$m: "; SGP_NO_INTERLEAVE=$m WSTATS_SHORT=1 timeout -k 10 200 python tools/wstats_copy_cost.py 2>&1 | grep "fresh arrays" | tail -1; done; done | tee $O/perpoint.txt
