#!/bin/bash
# round 4, GPU call AW: the driver's bench call (full bench.py --gpus 1 --steps 20 --warmup 5) with and without the interleaved enqueue, alternated
O=gpurun_out/r4aw; mkdir -p $O
for i in 1 2 3; do for m in 0 1; do SGP_NO_INTERLEAVE=$m timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > $O/b_$m_$i.json; python -c "
import json
d=json.loads(open('$O/b_$m_$i.json').read().strip().splitlines()[-1])
print('no_interleave=$m', round(d['value'],1), round(d['blocks']['ms_per_step_min']*1e3,1), round(d['ms_per_step']*1e3,1), round(d['blocks']['ms_per_step_max']*1e3,1), 'device', round(d['phases_us']['sweep_device'],1), 'per_point', [round(x['sweeps_per_s_with_w_stats']) for x in d['extra']['per_point']], 'C1', [round(c['sweeps_per_s']) for c in d['extra']['configs']][0])"; done; done | tee $O/driver_style_ab.txt
