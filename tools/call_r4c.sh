#!/bin/bash
# round 4, GPU call C: the overlapped order for data-sharded sweeps (one reduce per statistics group)
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?
tail -8 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/rehearse_two_ranks.py 2>&1 | grep -E "rank [01]\]|Error|error" > $O/rehearse_small.txt; tail -6 $O/rehearse_small.txt
RH_N=20000 RH_M=512 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/rehearse_two_ranks.py 2>&1 | grep -E "rank [01]\]|Error|error" > $O/rehearse_T.txt; tail -6 $O/rehearse_T.txt
SGP_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err; echo "2-rank bench rehearsal rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c/bench_2rank_rehearsal.json").read().strip().splitlines()[-1])
    print("2-rank rehearsal (gloo through the host; numbers mean nothing):", round(d["value"]), d["config"]["parallelism"], d["sweep_order"]["kind"])
    print("scaling leg:", json.dumps(d["extra"]["scaling_workload"])[:600])
    print("parity:", d["parity"])
except Exception as e:
    print("no bench line:", e)
PY
tail -5 $O/bench_2rank_rehearsal.err
bash tools/ab_multi.sh 2 "base|base|" "new|new|" 2>&1 | tee $O/ab.txt
echo done
