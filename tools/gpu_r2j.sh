#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest with rolled variant"; SINNERF_HIP_LIB=$PWD/build/variants/lib_dwrolled.so timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_grads_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "weight or grad or fp32 or oracle" > gpurun_out/pytest_j.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_j.log
for lib in sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_dwrolled.so; do
  echo "== $lib"; SINNERF_HIP_LIB=$PWD/$lib timeout 300 python tools/dw_time.py 2>&1 | grep -E "^tasks|^hot|^variant"
done | tee gpurun_out/dw_time_j.log
