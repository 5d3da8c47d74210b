#!/bin/bash
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
for mode in eager graph; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tt -o $mode -- python $R/tools/train_trace.py $mode > $R/gpurun_out/tt_$mode.log 2>&1
  echo "== $mode"; python $R/tools/train_trace.py --summarize $R/gpurun_out/tt/${mode}_kernel_trace.csv
done
