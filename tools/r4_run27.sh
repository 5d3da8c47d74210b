#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest (bf16 state kernels)"; timeout 1500 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_grads_gpu.py tests/test_bf16_configs_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_r27.log 2>&1; echo "exit $?"; tail -3 gpurun_out/pytest_r27.log
python tools/fwd_t_time.py 2>&1 | tail -2
python tools/chain_t_time.py 2>&1 | tail -2
bash tools/train_pmc.sh > gpurun_out/train_pmc.log 2>&1; grep "bf16_t_kernel" gpurun_out/train_pmc.txt
python tools/dp_leg_time.py 2>&1 | tail -3
