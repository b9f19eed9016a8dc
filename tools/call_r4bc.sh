#!/bin/bash
set -e
mkdir -p gpurun_out/r4bc
for i in 1 2; do
timeout -k 10 120 python tools/per_point_outliers.py nogc > gpurun_out/r4bc/outliers_nogc_$i.txt 2>&1
timeout -k 10 120 python tools/per_point_outliers.py > gpurun_out/r4bc/outliers_default_$i.txt 2>&1
done
grep -h -v amdgpu gpurun_out/r4bc/outliers_*_?.txt
