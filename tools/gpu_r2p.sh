#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 600 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_p.log 2>&1; echo "pytest exit $?"; grep -E "^FAILED|passed|failed" gpurun_out/pytest_p.log | tail -5
echo "== dw_time"; timeout 300 python tools/dw_time.py 2>&1 | grep -E "bf16 state" | tee gpurun_out/dw_time_p.log
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_p.log 2>&1; tail -1 gpurun_out/train_bench_p.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bf16_training']); print(d['ms_per_step'], d['fine_pass'])"
