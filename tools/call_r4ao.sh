#!/bin/bash
# round 4, GPU call AO: the per-point kernel's second workgroup of a CU started half a step late (s_sleep 32 x k = 0.85 us x k)
O=gpurun_out/r4ao; mkdir -p $O
D=gaussianprocessnode_amd/csrc
for v in fin st1 st2 st3 fin; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; echo -n "$v: "; timeout -k 10 100 python tools/quadform_alone.py 2>&1 | grep "quadform alone"; done | tee $O/stagger.txt
cp $D/libsgp_hip_fin.so $D/libsgp_hip.so
