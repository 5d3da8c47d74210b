#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/mlp_time.py sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_thin2.so build/variants/lib_thin4.so sinnerf_amd/csrc/libsinnerf_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mlp_time_i.log
