#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest (x3, grads, parity, classic, round3)"; timeout 1500 python -m pytest tests/test_bf16x3_gpu.py tests/test_grads_gpu.py tests/test_parity_gpu.py tests/test_classic_heads_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_r17.log 2>&1; echo "exit $?"; tail -5 gpurun_out/pytest_r17.log
