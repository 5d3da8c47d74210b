#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_f32noemb.so build/variants/lib_f32noacts.so build/variants/lib_f32nostore.so; do
  echo "$lib: $(SINNERF_HIP_LIB=$PWD/$lib timeout 120 python tools/fwd_train_time.py 2>&1 | tail -1)"
done | tee gpurun_out/fwd_train_ablation.log
