#!/bin/bash
R=$PWD; mkdir -p gpurun_out
for rep in 1 2; do
for v in cbase cap5 cap8 pf3 bg2 bg5; do
  printf "%-8s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_c_$v.so timeout 100 python tools/chain_t_time.py 2>&1 | grep kernel
done; done 2>&1 | tee gpurun_out/chain_knobs_ab.log
