#!/bin/bash
# A/B of two (or more) builds of csrc/libsgp_hip.so on the SAME GPU box, interleaved, three rounds: box-to-box variance is
# ~1.5 %, most kernel changes in round 1 were worth less.  Put the candidates at ab/lib_A.so, ab/lib_B.so, ... (ab/ is
# git-ignored but travels with gpurun), then:   gpurun --timeout 900 -- 'bash tools/ab_bench.sh A B'
L=gaussianprocessnode_amd/csrc/libsgp_hip.so
cp $L /tmp/ab_keep.so
for r in 1 2 3; do for v in "$@"; do cp ab/lib_$v.so $L; timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_us']; print('$v', round(d['value'],1), 'wall', round(d['ms_per_step']*1000,1), 'device', round(p['sweep_device'],1), 'F1', round(p['finish1_lambda_chain'],1), 'syrk', round(p['syrk'],1))"; done; done
cp /tmp/ab_keep.so $L
