#!/bin/bash
# round 4, GPU call D: the 16-wave SYRK (split-K continued inside the CU: a quarter of the slabs)
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -8 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 3 "narrow|new|" "wide|w16|" "wide_lib_narrow_env|w16|SGP_SYRK_WIDE=0" "quarter_slabs_experiment|qs|SGP_SYRK_WIDE=0" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_wide.txt 2>&1
SGP_SYRK_WIDE=0 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace_narrow.txt 2>&1
export HSA_ENABLE_IPC_MODE_LEGACY=0
RH_N=20000 RH_M=512 RH_D=8 RH_JIT=1e-6 RH_GTOL=1e-5 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/rehearse_two_ranks.py 2>&1 | grep -E "rank [01]\]|Error|error" > $O/rehearse_T.txt; tail -8 $O/rehearse_T.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_T.json 2> $O/bench_T.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4d/bench_T.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "launch_us", "launches_per_sweep")}, d["roofline"]["groups"])
print("single", d["roofline"]["single_launch_all_tiles_all_cus"])
PY
timeout -k 10 300 python bench.py --workload N1M --no-cpu-baseline > $O/bench_N1M.json 2> $O/bench_N1M.err; python -c "
import json; d=json.loads(open('gpurun_out/r4d/bench_N1M.json').read().strip().splitlines()[-1]); print('N1M', d['value'], d['roofline']['frac'], d['phases_us'])"
echo done
