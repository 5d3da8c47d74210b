#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "main                "; python tools/fwd_t_time.py 2>&1 | grep kernel
  for v in nt0 dmae dmae_nt0 novst_dmae novst; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/fwd_t_time.py 2>&1 | grep kernel
  done
done
} | tee gpurun_out/fwd_t_ablation3.log
