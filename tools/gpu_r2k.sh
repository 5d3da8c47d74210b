#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_bf16_configs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_k.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_k.log
bash tools/gpu_ab.sh sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_nointer.so
