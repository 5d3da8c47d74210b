#!/bin/bash
# One gpurun call: tests + bench + training bench + rocprof (kernel trace, then PMC passes).  Stage timeouts everywhere.
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench.log
echo "== train bench"; timeout 600 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1; echo "train exit $?"; tail -2 gpurun_out/train_bench.log
cd /tmp
echo "== rocprof stats"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1; echo "exit $?"
echo "== rocprof pmc1"; timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof -o pmc1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc1.log 2>&1; echo "exit $?"
echo "== rocprof pmc2"; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof -o pmc2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc2.log 2>&1; echo "exit $?"
echo "== rocprof pmc3"; timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof -o pmc3 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc3.log 2>&1; echo "exit $?"
timeout 60 rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
cd $R; ls -R gpurun_out/prof | head -30
