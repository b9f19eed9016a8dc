#!/bin/bash
# round 4, GPU call G: LDS-free K_uu Gram + K_uu chain gated on group 0's assembly; wide SYRK variants; config rates
O=gpurun_out/r4g; mkdir -p $O
D=gaussianprocessnode_amd/csrc
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "wide_only|w16g|" "cur2|cur2|" "cur2_kuu_gate_r3|cur2|SGP_KUU_ASM=0" "cur2_wide_g0_only|cur2|SGP_SYRK_WIDE=2" "cur2_cut4|cur2|SGP_OVERLAP_COLS=4" "cur2_cut25|cur2|SGP_OVERLAP_COLS=2,5" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_cur2.txt 2>&1
head -48 $O/sweep_trace_cur2.txt | grep -E "syrk|assemble|gram|Lambda step [0-8] |join_wait|K_uu step [08]|prep_xu|gemm32|trmv|scalars"
cp $D/libsgp_hip.so /tmp/keep_cur.so
for v in new cur2; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo "== config rates, $v"; timeout -k 10 200 python tools/config_rates.py 2>&1 | grep -v amdgpu | tee $O/config_rates_$v.txt; done
cp /tmp/keep_cur.so $D/libsgp_hip.so
echo done
