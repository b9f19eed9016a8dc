#!/bin/bash
O=gpurun_out/r4bs; mkdir -p $O
SGP_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
echo "rc $?"; tail -c 1500 $O/bench_2ranks.json; tail -5 $O/bench_2ranks.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4bs/bench_driver_style.json").read().strip().splitlines()[-1])
print("driver style", round(d["value"],1), d["blocks"], d["host_binding"], [ (p["config"], round(p["sweeps_per_s_with_w_stats"])) for p in d["extra"]["per_point"]])
PY
