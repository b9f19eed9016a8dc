#!/bin/bash
# round 4, GPU call AN: k_theta_grad_uf with the K_uf panel stored as it arrives; then the LDS counters of every kernel of the sweep
O=gpurun_out/r4an; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_tg1.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "theta or gradient or training or kin40k or streaming or device_paced or sharded" > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
for v in fin tg1 fin tg1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 120 python examples/train_kin40k.py 2>&1 | tail -1 | grep -o '"train_seconds": [0-9.]*'; done | tee $O/train.txt
cp $D/libsgp_hip_tg1.so $D/libsgp_hip.so
bash tools/call_r4am.sh
