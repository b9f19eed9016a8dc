#!/bin/bash
# round 4, GPU call Y: planner restricted (three groups only with short masked launches), mu staged with the tile in k_quadform_fused
O=gpurun_out/r4y; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_dir4.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
timeout -k 10 200 python tools/show_plans.py 2>&1 | grep -v amdgpu | tee $O/plans.txt
for v in dir3 dir4 dir3 dir4; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done | tee $O/wstats_time.txt
cp $D/libsgp_hip_dir4.so $D/libsgp_hip.so
for w in T C3 N100K N1M C2; do SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dir4 $w', round(d['value'],2))"; done | tee $O/rates.txt
