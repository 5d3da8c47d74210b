#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 600 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_o.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_o.log | tail -5
echo "== dw_time"; timeout 300 python tools/dw_time.py 2>&1 | grep -E "bf16" | tee gpurun_out/dw_time_o.log
