#!/bin/bash
# round-2 call A: tests + smoke + full bench line + kernel stats (bf16 inference, training)
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-600
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench.log
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof"
echo "== rocprof stats bf16"; timeout 300 $P --stats -o stats_bf16 -- python $R/bench.py --steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_stats_bf16.log 2>&1; echo "exit $?"
echo "== rocprof stats train"; timeout 300 $P --stats -o stats_train -- python $R/tools/train_bench.py > $R/gpurun_out/prof_stats_train.log 2>&1; echo "exit $?"
cd $R; ls gpurun_out/prof | head -40
