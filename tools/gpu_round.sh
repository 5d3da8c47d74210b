#!/bin/bash
# One gpurun call: tests + diagnostics + bench + rocprof.  Every stage under its own timeout.
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== diag"; timeout 600 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag exit $?"; tail -5 gpurun_out/diag.log
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -3 gpurun_out/bench.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1; echo "rocprof exit $?"; tail -3 $R/gpurun_out/prof.log
cd $R; ls -R gpurun_out/prof | head -20
