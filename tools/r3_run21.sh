#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  echo -n "main (2 WG x 80 KB)  "; python tools/dwn_time.py 2>&1 | grep total
  printf "%-21s" "dwn_1wg (128 KB)"; SINNERF_HIP_LIB=$R/build/variants/lib_dwn_1wg.so python tools/dwn_time.py 2>&1 | grep total
done
} | tee gpurun_out/dw_1wg_ab.log
