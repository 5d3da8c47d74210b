#!/bin/bash
R=$PWD; mkdir -p gpurun_out
for rep in 1 2 3; do
  printf "%-8s" main; timeout 100 python tools/step_time.py 2>&1 | grep step
  for v in costB costC costD costE; do
    printf "%-8s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so timeout 100 python tools/step_time.py 2>&1 | grep step
  done
done 2>&1 | tee gpurun_out/dw_cost_ab.log
