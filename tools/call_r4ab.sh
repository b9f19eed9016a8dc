#!/bin/bash
# round 4, GPU call AB: k_gram_uf as a pipelined loop over point blocks (large problems) against one workgroup per tile
O=gpurun_out/r4ab; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_gP.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
for v in cur gP cur gP; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; timeout -k 10 200 python tools/syrk_time.py $v 2>&1 | grep -v amdgpu | grep -o "^.*N=[0-9]*\|Gram.*" | paste - - ; done | tee $O/gram_time.txt
for v in cur gP; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; for w in T C3 N1M; do SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', round(d['value'],2))"; done; done | tee $O/rates.txt
cp $D/libsgp_hip_gP.so $D/libsgp_hip.so
