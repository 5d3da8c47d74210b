#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd /tmp
for rep in 1 2; do
for v in 5 7; do for q in quad noquad; do
  rm -rf /tmp/pf
  SINNERF_HIP_LIB=$R/build/variants/lib_only${v}_$q.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o t -- python $R/tools/dwn_time.py > /tmp/l.txt 2>&1 < /dev/null
  f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
  echo -n "variant $v $q: "; if [ -n "$f" ]; then grep "dw_narrow_bf16_asm" "$f" | cut -d, -f2-4; else echo none; fi
done; done; done 2>&1 | tee $R/gpurun_out/dwn_quad_per_shape.log
