#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_next_rows.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/dp_leg_time.py 2>&1 | grep ms/step | tee gpurun_out/dp_leg_time.log
