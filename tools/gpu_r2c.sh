#!/bin/bash
# round-2 call C: new weight-gradient entry + sink, bf16 kernels after packed softplus / prefetch, trunk ablations, training bench
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_grads_gpu.py tests/test_parity_gpu.py tests/test_bf16_configs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_c.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_c.log
echo "== mlp_time"; timeout 600 python tools/mlp_time.py sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_skip2.so build/variants/lib_nobar.so build/variants/lib_nodma.so build/variants/lib_nofrag.so build/variants/lib_noepi.so 2>&1 | tee gpurun_out/mlp_time_c.log
echo "== train bench"; timeout 300 python tools/train_bench.py > gpurun_out/train_bench_c.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench_c.log
