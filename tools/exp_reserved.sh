export SGP_SPIN_LIMIT=30000
for r in 2 1 3; do for c in "" "2,4" "3"; do
echo "== reserved=$r cols=$c"
SGP_RESERVED_PER_SE=$r SGP_OVERLAP_COLS=$c timeout -k 10 60 python tools/sweep_trace.py 2>&1 | grep -E "syrk|Lambda step|K_uu step 8|scalars" | awk '{printf "%s ", $0; print ""}' | cut -c1-60
SGP_RESERVED_PER_SE=$r SGP_OVERLAP_COLS=$c timeout -k 10 100 python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
d=json.loads(l[-1]); p=d['phases_us']; print('BENCH', round(d['value'],1), 'wall', round(d['ms_per_step']*1000,1), {k: round(v,1) for k,v in p.items()})"
done; done
