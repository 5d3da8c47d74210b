#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "main                "; python tools/fwd_t_time.py 2>&1 | grep kernel
  for v in spread1 spread0 novst novst0 nostage; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/fwd_t_time.py 2>&1 | grep kernel
  done
done
} | tee gpurun_out/fwd_t_ablation2.log
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "bit_for_bit" 2>&1 | tail -2
