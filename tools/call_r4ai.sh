#!/bin/bash
# round 4, GPU call AI: the chip-filling threshold at 10 000 (k_syrk_direct, gated K_uu chain and overlapped order for small problems too)
O=gpurun_out/r4ai; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_g10k.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
for v in fin g10k; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 120 python examples/train_kin40k.py 2>&1 | tail -1 | cut -c1-330; timeout -k 10 120 python examples/train_banana.py 2>&1 | tail -1 | cut -c1-260; done | tee $O/train.txt
cp $D/libsgp_hip_g10k.so $D/libsgp_hip.so
bash tools/ab_multi.sh 2 "fin|fin|" "g10k|g10k|" 2>&1 | tee $O/ab_T.txt
timeout -k 10 200 python tools/show_plans.py 2>&1 | grep -v amdgpu | tee $O/plans.txt
timeout -k 10 200 python tools/soak.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
