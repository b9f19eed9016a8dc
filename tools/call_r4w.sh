#!/bin/bash
# round 4, GPU call W: k_quadform_direct (per-point quadratic forms without LDS, one wave per item) against the LDS-staged kernel
O=gpurun_out/r4w; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip_qf1.so $D/libsgp_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; echo "pytest failed: stopping"; exit 1; fi
for v in dir3 qf1 dir3 qf1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo $v; timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu; done | tee $O/wstats_time.txt
for c in "3,5" "4" "5"; do echo "== C3 cut $c"; SGP_OVERLAP_COLS=$c timeout -k 10 120 python tools/sweep_trace.py 40000 512 8 2>&1 | grep -vE "amdgpu|workgroup exits|has its statistics"; done > $O/trace_C3.txt 2>&1
grep -E "==|syrk|assemble|Lambda step [0-8] |gram_uf" $O/trace_C3.txt
