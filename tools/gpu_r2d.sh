#!/bin/bash
# round-2 call D: transpose-read probe, bf16 dW with ds_read_b64_tr_b16, graph-captured training step
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tr probe"; hipcc --offload-arch=gfx950 -O2 -w tools/ubench/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe
echo "== pytest"; timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_grads_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_d.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/pytest_d.log
echo "== dw_time"; timeout 300 python tools/dw_time.py > gpurun_out/dw_time.log 2>&1; cat gpurun_out/dw_time.log | tail -40
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_d.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench_d.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_d.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v.get('ms_per_step'),v.get('error')) for k,v in d.items() if k.startswith('train')})"
