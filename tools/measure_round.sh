#!/bin/bash
# Every measurement behind profiles/r04_* and DESIGN.md's measured-state section, in TWO GPU calls (one call may run 20 minutes):
#   gpurun --timeout 1200 -- 'bash tools/measure_round.sh prof'     tests, the rocprofv3 passes (PMC counters, kernel trace + stats)
#   gpurun --timeout 1200 -- 'bash tools/measure_round.sh rest'     bench lines, in-kernel timelines, config rates, accuracy, training
# Outputs land under gpurun_out/final4/; copy what is judged into profiles/ (tools/collect_profiles.py).
# Under the profiler bench.py times ONE block (--blocks 1): the counter passes serialise the kernels, and 15 blocks of sweeps whose
# streams meet through device words took longer than the passes' 300 s limit.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final4; mkdir -p $O
cd $R
if [ "$1" = "prof" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
  h=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); cp $h $O/pmc_$c.csv; python3 $R/tools/pmc_summary.py $h $c > $O/pmc_$c.txt 2>&1; echo "pmc $c done"
done
python3 $R/tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv T $O/pmc_traffic.json; cp $O/pmc_traffic.json $R/profiles/r04_pmc_traffic.json
rm -rf /tmp/pmc_m && SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -- python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mfma.err
h=$(find /tmp/pmc_m -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_mfma_summary.py $h > $O/pmc_mfma.txt 2>&1; echo "pmc mfma done"
rm -rf /tmp/pmc_v && SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_v -- python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline > /dev/null 2> $O/pmc_valu.err
h=$(find /tmp/pmc_v -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_valu_summary.py $h > $O/pmc_valu.txt 2>&1; echo "pmc valu done"
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 60 --warmup 10 --blocks 1 --no-cpu-baseline > $O/kt.json 2> $O/kt.err
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); g=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 $R/tools/syrk_launches.py $f > $O/syrk_launches.txt 2>&1; cp $g $O/kernel_stats.csv; python3 $R/tools/kstats.py $g auto > $O/kernel_stats_per_sweep.txt 2>&1
echo "trace done"
# a hooked (data-sharded) device-paced training run under the kernel trace: one Gram and one SYRK launch per minibatch, no second pass
rm -rf /tmp/kh && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kh -- python3 $R/tools/hooked_train.py > $O/hooked_train.json 2> $O/hooked_train.err
echo "profiled hooked_train exit code: $?" > $O/hooked_train_rc.txt; cat $O/hooked_train_rc.txt
g=$(find /tmp/kh -name "*kernel_stats.csv" | head -1); python3 $R/tools/kstats.py $g auto > $O/hooked_train_kernel_stats.txt 2>&1; echo "hooked train done"
fi
if [ "$1" = "rest" ]; then
cd $R
timeout -k 10 300 python bench.py > $O/bench_T.json 2> $O/bench_T.err; tail -c 400 $O/bench_T.json; echo
SGP_OVERLAP=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_T_plain_order.json 2> /dev/null
timeout -k 10 300 python bench.py --workload N1M --no-cpu-baseline > $O/bench_N1M.json 2> $O/bench_N1M.err; tail -c 300 $O/bench_N1M.json; echo
SGP_TRACE_WGS=1 timeout -k 10 60 python tools/sweep_trace.py > $O/sweep_timeline_T.txt 2>&1
timeout -k 10 60 python tools/step_trace.py > $O/step_trace_T.txt 2>&1
SGP_OVERLAP=0 timeout -k 10 60 python tools/sweep_trace.py > $O/sweep_timeline_T_plain_order.txt 2>&1
timeout -k 10 60 python tools/sweep_trace.py 40000 512 8 20 > $O/sweep_timeline_C3.txt 2>&1
timeout -k 10 200 python tools/config_rates.py > $O/config_rates.txt 2>&1; tail -8 $O/config_rates.txt
timeout -k 10 300 python tests/scripts/accuracy_sweep.py 100 > $O/accuracy_sweep.txt 2>&1; grep "worst" $O/accuracy_sweep.txt
timeout -k 10 120 python examples/train_kin40k.py > $O/train_kin40k.txt 2>&1; tail -2 $O/train_kin40k.txt
timeout -k 10 120 python examples/train_banana.py > $O/train_banana.txt 2>&1; tail -2 $O/train_banana.txt
timeout -k 10 200 python tools/soak.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/rehearse_two_ranks.py 2>&1 | grep -E "rank [01]\]" > $O/rehearse_two_ranks.txt; tail -3 $O/rehearse_two_ranks.txt
RH_N=20000 RH_M=512 RH_D=8 RH_JIT=1e-6 RH_GTOL=1e-5 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/rehearse_two_ranks.py 2>&1 | grep -E "rank [01]\]" > $O/rehearse_two_ranks_overlapped.txt; tail -3 $O/rehearse_two_ranks_overlapped.txt
timeout -k 10 200 python tools/wstats_time.py 2>&1 | grep -v amdgpu > $O/wstats_time.txt; cat $O/wstats_time.txt
SGP_SYRK_WIDE=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_T_syrk_256_threads.json 2> /dev/null
# same-box A/B of library builds (tools/ab_multi.sh): round 3's (commit 0f7549a), the first half of round 4 (8dfa5c6: LDS-staged 16-wave SYRK,
# one cut) and this one.  The old libraries are built by hand from `git show <commit>:...`; a library that lacks an
# export bench.py now calls prints no line.
D=gaussianprocessnode_amd/csrc
if [ -f $D/libsgp_hip_r4a.so ]; then
cp $D/libsgp_hip.so $D/libsgp_hip_fin.so; bash tools/ab_multi.sh 3 "round3_library|base|" "round4_first_half|r4a|" "round4_final|fin|" 2>&1 | grep -v Traceback | grep -v "^  File\|IndexError\|^    " > $O/ab_r3_vs_r4.txt; cat $O/ab_r3_vs_r4.txt
for w in C3 N1M; do for v in r4a fin; do cp $D/libsgp_hip_$v.so $D/libsgp_hip.so; SGP_BENCH_SKIP_ALONE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', round(d['value'],2), 'sweeps/s')"; done; done >> $O/ab_r3_vs_r4.txt 2>&1; tail -4 $O/ab_r3_vs_r4.txt
cp $D/libsgp_hip_fin.so $D/libsgp_hip.so
fi
timeout -k 10 60 ./tools/mfma_f64_probe > $O/mfma_f64_probe.txt 2>&1; timeout -k 10 60 ./tools/dpp_f64_probe > $O/dpp_f64_probe.txt 2>&1; timeout -k 10 120 ./tools/syrk_direct_probe > $O/syrk_direct_probe.txt 2>&1; timeout -k 10 60 ./tools/store_bw_probe > $O/store_bw_probe.txt 2>&1
echo "measure_round done"

fi
