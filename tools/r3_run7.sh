#!/bin/bash
# round 3, GPU call 7: full GPU suite + smoke + bench with the hand-scheduled bf16 training kernels; PMC of the training kernels
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-300
echo "== train pmc"; bash tools/train_pmc.sh > gpurun_out/train_pmc.log 2>&1; grep "bf16" gpurun_out/train_pmc.txt
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof"
timeout 600 $P --stats -o stats_train -- python $R/tools/train_bench.py > $R/gpurun_out/prof_stats_train.log 2>&1; echo "stats exit $?"
timeout 600 $P --pmc FETCH_SIZE -o pmc_train_fetch -- python $R/tools/train_bench.py > $R/gpurun_out/prof_pmc_train_fetch.log 2>&1; echo "exit $?"
timeout 600 $P --pmc WRITE_SIZE -o pmc_train_write -- python $R/tools/train_bench.py > $R/gpurun_out/prof_pmc_train_write.log 2>&1; echo "exit $?"
