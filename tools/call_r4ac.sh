#!/bin/bash
# round 4, GPU call AC: LDS operands of the tile products requested a k-step ahead (tile_mma, the step kernel's own-tile update)
O=gpurun_out/r4ac; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_mm1.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "fin|fin|" "mm1|mm1|" 2>&1 | tee $O/ab_T.txt
for w in C2 C3; do EXTRA_ARGS="--workload $w" STEPS=300 bash tools/ab_multi.sh 2 "fin_$w|fin|" "mm1_$w|mm1|"; done 2>&1 | tee $O/ab_other.txt
for v in fin mm1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu; done | tee $O/config_rates.txt
cp $D/libsgp_hip_mm1.so $D/libsgp_hip.so
