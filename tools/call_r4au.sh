#!/bin/bash
# round 4, GPU call AU: the two chains' launches enqueued alternately (one-shot sweeps)
O=gpurun_out/r4au; mkdir -p $O
D=gaussianprocessnode_amd/csrc
V=${1:-il1}
cp $D/libsgp_hip_$V.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
STEPS=20 bash tools/ab_multi.sh 3 "fin-steps20|fin|" "$V-steps20|$V|" 2>&1 | tee $O/ab_T_steps20.txt
bash tools/ab_multi.sh 2 "fin|fin|" "$V|$V|" 2>&1 | tee $O/ab_T.txt
for v in fin $V fin $V; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo -n "$v: "; WSTATS_SHORT=1 timeout -k 10 200 python tools/wstats_copy_cost.py 2>&1 | grep "fresh arrays" | tail -1; done | tee $O/perpoint.txt
for w in C2 C3; do EXTRA_ARGS="--workload $w" STEPS=20 bash tools/ab_multi.sh 2 "fin_$w|fin|" "${V}_$w|$V|"; done 2>&1 | tee $O/ab_other.txt
cp $D/libsgp_hip_$V.so $D/libsgp_hip.so
timeout -k 10 200 python tools/soak.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
