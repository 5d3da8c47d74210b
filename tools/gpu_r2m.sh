#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 900 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_bf16_configs_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_m.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_m.log | tail -8
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_m.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench_m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bf16_training']); print(d['ms_per_step'])"
