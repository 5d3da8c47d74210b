#!/bin/bash
# round 3, GPU call 3: timing ablations of the hand-scheduled training forward (kernel-only, fine pass)
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "main                "; python tools/fwd_t_time.py 2>&1 | grep kernel
  echo -n "compiler-scheduled  "; SINNERF_COMPILER_SCHEDULED=1 python tools/fwd_t_time.py 2>&1 | grep kernel
  for v in base nostage nosign cap5 cap8 pf3 bar6 noemb notrunk; do
    printf "%-20s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/fwd_t_time.py 2>&1 | grep kernel
  done
done
} | tee gpurun_out/fwd_t_ablation.log
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "llff" 2>&1 | tail -2
