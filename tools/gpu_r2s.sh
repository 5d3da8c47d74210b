#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 900 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_bf16_configs_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_s.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_s.log | tail -6
echo "== stages: $(timeout 120 python tools/bf16_stage_time.py 2>&1 | tail -1)"
echo "== fwd_train fp32: $(timeout 120 python tools/fwd_train_time.py 2>&1 | tail -1)"
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_s.log 2>&1; tail -1 gpurun_out/train_bench_s.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bf16_training']); print(d['ms_per_step'], d['fine_pass']['fwd_train_ms'])"
