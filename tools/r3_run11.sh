#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "fwd main            "; python tools/fwd_t_time.py 2>&1 | grep kernel
  for v in f_norays f_norays_noemb f_nodirst; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so timeout 120 python tools/fwd_t_time.py 2>&1 | grep kernel
  done
done
} | tee gpurun_out/fwd_t_ablation4.log
