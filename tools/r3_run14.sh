#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
{
for rep in 1 2; do
  echo -n "main (80 KB)   "; python tools/bf16_stage_time.py 2>&1 | grep "S="
  for v in dwn_65536 dwn_73728; do printf "%-15s" $v; SINNERF_HIP_LIB=$R/build/variants/lib_$v.so python tools/bf16_stage_time.py 2>&1 | grep "S="; done
done
} | tee gpurun_out/dw_narrow_ab.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dwn -o st -- python $R/tools/train_bench.py > $R/gpurun_out/dwn_train.log 2>&1
cd $R
grep "dw_\|fwd_bf16_t\|chain_bf16_t" gpurun_out/dwn/st_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
tail -1 gpurun_out/dwn_train.log | cut -c1-400
