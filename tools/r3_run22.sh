#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  echo -n "main (2 WG x 80 KB)  "; python tools/dwn_time.py 2>&1 | grep total
  printf "%-21s" "dwn_3wg (3 x 52 KB)"; SINNERF_HIP_LIB=$R/build/variants/lib_dwn_3wg.so python tools/dwn_time.py 2>&1 | grep total
done
} | tee gpurun_out/dw_3wg_ab.log
SINNERF_HIP_LIB=$R/build/variants/lib_dwn_3wg.so timeout 300 python -m pytest tests/test_grads_gpu.py -m gpu -q -p no:cacheprovider -k "weight_grad or golden" 2>&1 | tail -2
