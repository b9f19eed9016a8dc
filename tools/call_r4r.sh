#!/bin/bash
# round 4, GPU call R: what bounds k_syrk_stream16's steady state?  Timing variants (wrong results) against the product kernel
O=gpurun_out/r4r; mkdir -p $O
D=gaussianprocessnode_amd/csrc
cp $D/libsgp_hip.so /tmp/keep.so
for v in pf1 xNOSYNC xNOLOAD xNOSTORE xALL pf1; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; timeout -k 10 200 python tools/syrk_time.py $v 2>&1 | grep -v amdgpu; done | tee $O/syrk_variants.txt
cp /tmp/keep.so $D/libsgp_hip.so
