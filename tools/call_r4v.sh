#!/bin/bash
# round 4, GPU call V: the refitted overlap planner (one or two cuts) against forced plans at T, C3, N = 100 000, C2
O=gpurun_out/r4v; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_dir3.so $D/libsgp_hip.so
timeout -k 10 200 python tools/show_plans.py 2>&1 | grep -v amdgpu | tee $O/plans.txt
bash tools/ab_multi.sh 2 "auto|dir3|" "cut2,3|dir3|SGP_OVERLAP_COLS=2,3" "cut2,5|dir3|SGP_OVERLAP_COLS=2,5" "cut2,6|dir3|SGP_OVERLAP_COLS=2,6" "cut1,3|dir3|SGP_OVERLAP_COLS=1,3" "cut3|dir3|SGP_OVERLAP_COLS=3" 2>&1 | tee $O/ab_T.txt
EXTRA_ARGS="--workload C3" STEPS=300 bash tools/ab_multi.sh 2 "auto|dir3|" "cut5|dir3|SGP_OVERLAP_COLS=5" "cut4|dir3|SGP_OVERLAP_COLS=4" "cut3,5|dir3|SGP_OVERLAP_COLS=3,5" "cut4,6|dir3|SGP_OVERLAP_COLS=4,6" 2>&1 | tee $O/ab_C3.txt
EXTRA_ARGS="--workload N100K" STEPS=200 bash tools/ab_multi.sh 2 "auto|dir3|" "cut5|dir3|SGP_OVERLAP_COLS=5" "cut5,7|dir3|SGP_OVERLAP_COLS=5,7" "plain|dir3|SGP_OVERLAP=0" 2>&1 | tee $O/ab_N100K.txt
EXTRA_ARGS="--workload C2" STEPS=1000 bash tools/ab_multi.sh 2 "auto|dir3|" "cut1|dir3|SGP_OVERLAP_COLS=1" "cut2|dir3|SGP_OVERLAP_COLS=2" "cut1,2|dir3|SGP_OVERLAP_COLS=1,2" "plain|dir3|SGP_OVERLAP=0" 2>&1 | tee $O/ab_C2.txt
