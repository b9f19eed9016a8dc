#!/bin/bash
set -e
mkdir -p gpurun_out/r4bb
timeout -k 10 240 python tools/per_point_loop_split.py > gpurun_out/r4bb/split.txt 2>&1
cat gpurun_out/r4bb/split.txt
