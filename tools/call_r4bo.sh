#!/bin/bash
set -e
O=gpurun_out/r4bo; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_sweep_at_a_time" 2>&1 | tail -15
