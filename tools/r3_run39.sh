#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd /tmp
for v in 2 4 5 6 7; do
  rm -rf /tmp/pf$v
  SINNERF_HIP_LIB=$R/build/variants/lib_only$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$v -o t -- python $R/tools/dwn_time.py > /tmp/l$v.txt 2>&1 < /dev/null
  f=$(find /tmp/pf$v -name "*kernel_stats.csv" | head -1)
  echo -n "variant $v: "; if [ -n "$f" ]; then grep "dw_narrow_bf16_asm" "$f" | cut -d, -f1-4; else echo none; fi
done 2>&1 | tee $R/gpurun_out/dw_narrow_per_variant.log
