#!/bin/bash
# round 4, GPU call AA: what bounds k_gram_uf (3.2 TB/s of stores where plain stores reach 6.7)?  Timing variants (wrong results)
O=gpurun_out/r4aa; mkdir -p $O
D=gaussianprocessnode_amd/csrc
for v in cur gL cur gL; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; timeout -k 10 200 python tools/syrk_time.py $v 2>&1 | grep -v amdgpu | grep -o "^.*N=[0-9]*\|Gram.*" | paste - - ; done | tee $O/gram_variants.txt
cp $D/libsgp_hip_cur.so $D/libsgp_hip.so
