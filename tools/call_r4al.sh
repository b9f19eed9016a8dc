#!/bin/bash
# round 4, GPU call AL: k_gram_uf one point at a time (stores start after a quarter of the arithmetic)
O=gpurun_out/r4al; mkdir -p $O
D=gaussianprocessnode_amd/csrc
V=${1:-gr1}
for v in fin $V fin $V; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; timeout -k 10 200 python tools/syrk_time.py $v 2>&1 | grep -v amdgpu | grep -o "^.*N=[0-9]*\|Gram.*" | paste - - ; done | tee $O/gram_time.txt
bash tools/ab_multi.sh 3 "fin|fin|" "$V|$V|" 2>&1 | tee $O/ab_T.txt
for w in C2 N1M; do EXTRA_ARGS="--workload $w" STEPS=100 bash tools/ab_multi.sh 1 "fin_$w|fin|" "${V}_$w|$V|"; done 2>&1 | tee $O/ab_other.txt
cp $D/libsgp_hip_$V.so $D/libsgp_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
