#!/bin/bash
R=$PWD; mkdir -p gpurun_out
for rep in 1 2 3; do
for v in base dph5 dph18; do
  printf "%-8s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so timeout 100 python tools/fwd_t_time.py 2>&1 | grep kernel
done; done 2>&1 | tee gpurun_out/fwd_dephase_ab.log
timeout 200 python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "training_pack" 2>&1 | tail -2
