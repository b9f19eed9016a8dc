#!/bin/bash
O=gpurun_out/r4br; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
STEPS=20 bash tools/ab_multi.sh 4 "chain_after_chain|new|SGP_INTERLEAVE=0" "alternating|new|SGP_X=1" > $O/ab_steps20.txt 2>&1; cat $O/ab_steps20.txt
bash tools/ab_multi.sh 3 "chain_after_chain|new|SGP_INTERLEAVE=0" "alternating|new|SGP_X=1" > $O/ab_steps1000.txt 2>&1; cat $O/ab_steps1000.txt
timeout -k 10 200 python tools/soak.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
