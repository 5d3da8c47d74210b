#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 900 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_h.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_h.log
echo "== dw_time"; timeout 300 python tools/dw_time.py > gpurun_out/dw_time.log 2>&1; grep -v "amdgpu.ids" gpurun_out/dw_time.log | head -20
