#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "fwd main            "; python tools/fwd_t_time.py 2>&1 | grep kernel
  for v in f_novst f_nostage f_bare f_bare_noemb f_notrunk; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so timeout 120 python tools/fwd_t_time.py 2>&1 | grep kernel
  done
  echo -n "chain main          "; python tools/chain_t_time.py 2>&1 | grep kernel
  for v in c_novst c_nostage c_bare c_cap8 c_cap5; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so timeout 120 python tools/chain_t_time.py 2>&1 | grep kernel
  done
done
} | tee gpurun_out/floor_ablation.log
