#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd /tmp
for tag in asm compiler; do
  if [ $tag = compiler ]; then export SINNERF_DW_NARROW_COMPILER=1; fi
  rm -rf /tmp/prof_$tag
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $R/tools/dwn_time.py > /tmp/log_$tag.txt 2>&1 < /dev/null
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; grep total /tmp/log_$tag.txt
  if [ -n "$f" ]; then head -8 "$f" | cut -c1-200; else echo "no stats file"; ls -R /tmp/prof_$tag | head; fi
done 2>&1 | tee $R/gpurun_out/dwn_asm_prof.log
