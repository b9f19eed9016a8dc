# A/B of the overlapped sweep's knobs on one box: in-kernel timeline (tools/sweep_trace.py) + bench line per configuration
#   gpurun -- 'bash tools/exp_overlap.sh "1 1 3" "1 1 2,4" ... > gpurun_out/exp.txt'      (overlap g1_after cols)
export SGP_SPIN_LIMIT=${SGP_SPIN_LIMIT:-30000}
for cfg in "$@"; do
  set -- $cfg
  echo "== overlap=$1 g1_after=$2 cols=$3"
  SGP_OVERLAP=$1 SGP_G1_AFTER=$2 SGP_OVERLAP_COLS=$3 timeout -k 10 60 python tools/sweep_trace.py 2>&1 | grep -v amdgpu.ids | tail -52
  SGP_OVERLAP=$1 SGP_G1_AFTER=$2 SGP_OVERLAP_COLS=$3 timeout -k 10 100 python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
if l:
    d=json.loads(l[-1]); p=d['phases_us']; print('BENCH', round(d['value'],1), 'wall', round(d['ms_per_step']*1000,1), {k: round(v,1) for k,v in p.items()})
else: print('BENCH failed')"
done
