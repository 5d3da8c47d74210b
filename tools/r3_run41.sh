#!/bin/bash
R=$PWD; mkdir -p gpurun_out
{
timeout 200 python tools/dwn_cmp.py /tmp/new.pt
SINNERF_HIP_LIB=$R/build/variants/lib_noquad.so timeout 200 python tools/dwn_cmp.py /tmp/old.pt
timeout 100 python tools/dwn_cmp.py /tmp/old.pt /tmp/new.pt
for rep in 1 2 3; do
  echo -n "quad    "; timeout 100 python tools/step_time.py 2>&1 | grep step
  echo -n "pairs   "; SINNERF_HIP_LIB=$R/build/variants/lib_noquad.so timeout 100 python tools/step_time.py 2>&1 | grep step
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dwn_quad_ab.log
