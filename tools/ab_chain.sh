#!/bin/bash
# time the fp32 backward chain of several builds on one box: tools/ab_chain.sh variant.so ...  (log: gpurun_out/ab_chain.log)
mkdir -p gpurun_out
{
python -m pytest tests/test_grads_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for lib in "" "$@"; do
  echo -n "lib=${lib:-main}  "
  SINNERF_HIP_LIB=$lib python tools/chain_time.py 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/ab_chain.log
