#!/bin/bash
O=gpurun_out/final4; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_T.json 2> $O/bench_T.err; tail -c 300 $O/bench_T.json; echo
timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu > $O/wstats_time.txt; cat $O/wstats_time.txt
timeout -k 10 200 python tools/config_rates.py > $O/config_rates.txt 2>&1; tail -6 $O/config_rates.txt
SGP_TRACE_WGS=1 timeout -k 10 60 python tools/sweep_trace.py > $O/sweep_timeline_T.txt 2>&1
