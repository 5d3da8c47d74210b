#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 200 python tools/dwn_cmp.py /tmp/new.pt
SINNERF_DW_NARROW_COMPILER=1 timeout 200 python tools/dwn_cmp.py /tmp/old.pt
timeout 100 python tools/dwn_cmp.py /tmp/old.pt /tmp/new.pt
for rep in 1 2 3; do
  echo -n "generated narrow   "; timeout 120 python tools/dwn_time.py 2>&1 | grep total
  echo -n "compiler narrow    "; SINNERF_DW_NARROW_COMPILER=1 timeout 120 python tools/dwn_time.py 2>&1 | grep total
done
} 2>&1 | tee gpurun_out/dwn_asm_ab.log
timeout 400 python -m pytest tests/test_grads_gpu.py -m gpu -q -p no:cacheprovider -k "weight_grad or golden" 2>&1 | tail -3
