#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
  echo -n "main        "; python tools/dwn_time.py 2>&1 | grep total
  for v in dwc_bytes dwc_a dwc_b; do printf "%-12s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/dwn_time.py 2>&1 | grep total; done
done
} | tee gpurun_out/dw_cost_ab.log
