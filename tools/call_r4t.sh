#!/bin/bash
# round 4, GPU call T: k_syrk_direct (no LDS staging, a whole tile per wave) in the library against the LDS-staged 16-wave kernel
O=gpurun_out/r4t; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_dir1.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then echo "pytest failed: stopping"; exit 1; fi
bash tools/ab_multi.sh 2 "pf1|pf1|" "dir1|dir1|" 2>&1 | tee $O/ab_T.txt
for v in pf1 dir1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; timeout -k 10 200 python tools/syrk_time.py $v 2>&1 | grep -v amdgpu; done | tee $O/syrk_time.txt
cp $D/libsgp_hip_dir1.so $D/libsgp_hip.so
for w in N1M C3; do SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dir1 $w', round(d['value'],2))"; done | tee $O/other.txt
