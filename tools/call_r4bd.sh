#!/bin/bash
set -e
mkdir -p gpurun_out/final4
timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu > gpurun_out/final4/wstats_time.txt; cat gpurun_out/final4/wstats_time.txt
timeout -k 10 300 python bench.py > gpurun_out/final4/bench_T.json 2> gpurun_out/final4/bench_T.err; tail -c 300 gpurun_out/final4/bench_T.json
