#!/bin/bash
# round 4, GPU call AQ: when may the masked groups' first SYRK start?  SGP_G1_AFTER = 2 (default: when group 0's assembly starts) / 0 (when group 0's
# SYRK is resident) / 1 (when group 0 is assembled) -- remeasured with k_syrk_direct
O=gpurun_out/r4aq; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip.so $D/libsgp_hip_fin.so
bash tools/ab_multi.sh 3 "after2|fin|" "after0|fin|SGP_G1_AFTER=0" "after1|fin|SGP_G1_AFTER=1" 2>&1 | tee $O/ab_T.txt
EXTRA_ARGS="--workload C3" STEPS=200 bash tools/ab_multi.sh 2 "C3-after2|fin|" "C3-after0|fin|SGP_G1_AFTER=0" 2>&1 | tee $O/ab_C3.txt
EXTRA_ARGS="--workload C2" STEPS=500 bash tools/ab_multi.sh 2 "C2-after2|fin|" "C2-after0|fin|SGP_G1_AFTER=0" 2>&1 | tee $O/ab_C2.txt
