#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_grads_gpu.py tests/test_bf16_configs_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -12
{
for rep in 1 2; do
  echo -n "chain hand-scheduled      "; python tools/chain_t_time.py 2>&1 | grep kernel
  echo -n "chain compiler-scheduled  "; SINNERF_COMPILER_SCHEDULED=1 python tools/chain_t_time.py 2>&1 | grep kernel
  echo -n "fwd hand-scheduled        "; python tools/fwd_t_time.py 2>&1 | grep kernel
done
} | tee gpurun_out/chain_t_ab.log
python tools/train_bench.py 2>&1 | tail -3
