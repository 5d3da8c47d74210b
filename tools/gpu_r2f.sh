#!/bin/bash
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_grads_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_f.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/pytest_f.log
echo "== dw_time"; timeout 300 python tools/dw_time.py > gpurun_out/dw_time.log 2>&1; grep "bf16 state" gpurun_out/dw_time.log
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_f.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench_f.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_f.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v.get('ms_per_step'),v.get('error')) for k,v in d.items() if k.startswith('train')})"
