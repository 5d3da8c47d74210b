#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "main        "; python tools/dwn_time.py 2>&1 | grep total
  for v in dwsy6 dwsy4; do printf "%-12s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/dwn_time.py 2>&1 | grep total; done
done
} | tee gpurun_out/dw_sy_ab.log
timeout 300 python -m pytest tests/test_grads_gpu.py -m gpu -q -p no:cacheprovider -k "weight_grad" 2>&1 | tail -2
