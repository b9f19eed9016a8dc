#!/bin/bash
# round 4, GPU call B: tests on the new build (fused per-point kernel, exit handlers, sticky status word, large-M slabs), the
# profiled hooked training run that crashed at exit in round 3, A/B against the round-3 library, per-point rates
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; rc=$?
tail -15 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 2 "base|base|" "new|new|" 2>&1 | tee $O/ab.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_T.json 2> $O/bench_T.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4b/bench_T.json").read().strip().splitlines()[-1])
print("value", d["value"], "blocks", d["blocks"])
print("per_point", json.dumps(d["extra"]["per_point"], indent=1)[:1500])
print("configs", [(c["config"][:8], round(c["sweeps_per_s"])) for c in d["extra"]["configs"]])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kh && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kh -- python3 $GRAFT_REPO_ROOT/tools/hooked_train.py > $GRAFT_REPO_ROOT/$O/hooked_train.json 2> $GRAFT_REPO_ROOT/$O/hooked_train.err; echo "profiled hooked_train exit code: $?" | tee $GRAFT_REPO_ROOT/$O/hooked_train_rc.txt
cd $GRAFT_REPO_ROOT
tail -3 $O/hooked_train.err
echo done
