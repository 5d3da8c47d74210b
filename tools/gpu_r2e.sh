#!/bin/bash
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_grads_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_e.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/pytest_e.log
echo "== dw_time"; timeout 300 python tools/dw_time.py > gpurun_out/dw_time.log 2>&1; cat gpurun_out/dw_time.log | tail -40
