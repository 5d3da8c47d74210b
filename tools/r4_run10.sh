#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "weight_grad or gradients_golden or chain" > gpurun_out/pytest_bf16x3.log 2>&1; echo "exit $?"; grep -E "bf16x3|passed|failed|Error|error|assert" gpurun_out/pytest_bf16x3.log | head -20
bash tools/r4_run5.sh 2>&1 | grep -E "ms / step|dw_|mlp_"
bash tools/r4_run9.sh 2>&1 | tail -6
