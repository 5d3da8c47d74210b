#!/bin/bash
# A/B of fp32 training kernels on one box: tools/ab_train_f32.sh [variant.so ...]  (log: gpurun_out/ab_train_f32.log)
mkdir -p gpurun_out
{
python -m pytest tests/test_grads_gpu.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
for lib in "" "$@"; do
  echo "== lib=${lib:-main}"
  SINNERF_HIP_LIB=$lib python tools/chain_time.py 2>&1 | grep -v amdgpu.ids
  SINNERF_HIP_LIB=$lib python tools/fwd_train_time.py 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/ab_train_f32.log
