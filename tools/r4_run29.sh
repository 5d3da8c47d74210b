#!/bin/bash
export R=$PWD
for lib in build/variants/lib_968029c.so build/variants/lib_63ece53.so build/variants/lib_9f23e12.so ""; do
  echo "== ${lib:-in-tree}"
  SINNERF_HIP_LIB=${lib:+$PWD/$lib} python tools/x3_determinism.py 2>&1 | grep -a "^bf16x3\|autograd rgb" | cut -c1-200
done
