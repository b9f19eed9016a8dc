#!/bin/bash
# round 4, GPU call AG: from which size is the SYRK "chip-filling" (K_uu chain gated behind it, k_syrk_direct, overlapped order)?  SGP_GATE_MIN
O=gpurun_out/r4ag; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip.so $D/libsgp_hip_fin.so
for w in C2 N5K; do EXTRA_ARGS="--workload $w" STEPS=500 bash tools/ab_multi.sh 2 "$w-gate200k|fin|" "$w-gate100k|fin|SGP_GATE_MIN=100000" "$w-gate50k|fin|SGP_GATE_MIN=50000" "$w-gate100k-plain|fin|SGP_GATE_MIN=100000 SGP_OVERLAP=0"; done 2>&1 | tee $O/ab_gate.txt
SGP_GATE_MIN=100000 timeout -k 10 100 python tools/show_plans.py 2>&1 | grep -v amdgpu | tee $O/plans_gate100k.txt
