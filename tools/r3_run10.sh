#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "bit_for_bit" 2>&1 | tail -3
{
for rep in 1 2; do
  echo -n "chain main          "; python tools/chain_t_time.py 2>&1 | grep kernel
  echo -n "chain novst         "; SINNERF_HIP_LIB=$R/build/variants/lib_c_novst2.so python tools/chain_t_time.py 2>&1 | grep kernel
  echo -n "chain compiler-sch  "; SINNERF_COMPILER_SCHEDULED=1 python tools/chain_t_time.py 2>&1 | grep kernel
  echo -n "fwd main            "; python tools/fwd_t_time.py 2>&1 | grep kernel
done
} | tee gpurun_out/chain_t_ab2.log
