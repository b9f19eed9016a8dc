#!/bin/bash
# round 4, GPU call Q: SYRK with the omega weighting moved behind the products (loads no longer waited for in front of the MFMAs)
O=gpurun_out/r4q; mkdir -p $O
D=gaussianprocessnode_amd/csrc
bash tools/ab_multi.sh 2 "head|head|" "pf1|pf1|" 2>&1 | tee $O/ab_T.txt
for v in head pf1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so
  SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload N1M --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v N1M', round(d['value'],2), d['roofline'])"; done | tee $O/ab_N1M.txt
cp $D/libsgp_hip_pf1.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
