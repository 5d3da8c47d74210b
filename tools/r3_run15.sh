#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
{
for rep in 1 2; do
  echo -n "main (fp32 narrow 2 WG/CU)  "; python tools/train_bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 step %.3f ms  dW %.3f ms   bf16 step %.3f  dW %.3f' % (d['ms_per_step'], d['fine_pass']['dW_gemm_ms'], d['bf16_training']['ms_per_step'], d['bf16_training']['fine_dW_ms']))"
  echo -n "dwf32_1wg                   "; SINNERF_HIP_LIB=$R/build/variants/lib_dwf32_1wg.so python tools/train_bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 step %.3f ms  dW %.3f ms   bf16 step %.3f  dW %.3f' % (d['ms_per_step'], d['fine_pass']['dW_gemm_ms'], d['bf16_training']['ms_per_step'], d['bf16_training']['fine_dW_ms']))"
done
} | tee gpurun_out/dw_f32_narrow_ab.log
