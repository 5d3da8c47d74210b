#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest (all gpu tests except convergence)"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_convergence_gpu.py > gpurun_out/pytest_r23.log 2>&1; echo "exit $?"; tail -3 gpurun_out/pytest_r23.log
echo "== convergence"; SN_CONV_STEPS=2000 timeout 900 python -m pytest tests/test_convergence_gpu.py -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_conv.log 2>&1; echo "exit $?"; grep -a "mean_final_psnr" gpurun_out/pytest_conv.log | tail -1 | cut -c1-600; tail -1 gpurun_out/pytest_conv.log
