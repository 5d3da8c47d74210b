#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_chainnost.so build/variants/lib_fwdnost.so; do
  echo "$lib: $(SINNERF_HIP_LIB=$PWD/$lib timeout 120 python tools/bf16_stage_time.py 2>&1 | tail -1)"
done | tee gpurun_out/bf16_stage_ablation.log
