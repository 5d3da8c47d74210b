#!/bin/bash
R=$PWD; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_grads_gpu.py tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do timeout 100 python tools/step_time.py --graph 2>&1 | grep step | tr '\n' ' '; echo; done | tee gpurun_out/step_time_final.log
