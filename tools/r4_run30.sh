#!/bin/bash
export R=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
python tools/x3_determinism.py 2>&1 | grep -a "^bf16x3\|^fp32\|autograd rgb\|no_grad  rgb" | cut -c1-220
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|llff-patch\|all-fp32\|Error" | tail -6
python tools/x3_step_time.py bf16x3 2>&1 | tail -1
