#!/bin/bash
# round 4, GPU call A: tests on the new build, then same-box A/B of the launch-overhead experiments
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed: stopping"; exit 1; fi
bash tools/ab_multi.sh 2 "base|base|" "new|new|" "new_gatefirst|new|SGP_KUU_EARLY=0" "kpre|kpre|" "new_devkarg1|new|HIP_FORCE_DEV_KERNARG=1" "new_devkarg0|new|HIP_FORCE_DEV_KERNARG=0" 2>&1 | tee $O/ab.txt
SGP_TRACE_WGS=1 timeout -k 10 120 python tools/sweep_trace.py > $O/sweep_trace.txt 2>&1
timeout -k 10 120 python tools/step_trace.py > $O/step_trace.txt 2>&1
echo done
