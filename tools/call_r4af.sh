#!/bin/bash
# round 4, GPU call AF: k_quadform_fused with the XCD-aware block map
O=gpurun_out/r4af; mkdir -p $O
D=gaussianprocessnode_amd/csrc
V=${1:-qx1}
cp $D/libsgp_hip_$V.so $D/libsgp_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rules.py -m gpu -q -x -k "sweep_matches_oracle or direct_syrk or rule or cold or uncertain or w_stats or probit or classification" > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
for v in fin $V fin $V; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done | tee $O/wstats_time_$V.txt
cp $D/libsgp_hip_$V.so $D/libsgp_hip.so
