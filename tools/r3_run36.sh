#!/bin/bash
R=$PWD; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_grads_gpu.py tests/test_round2_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o t -- python $R/tools/step_time.py > /tmp/st.log 2>&1 < /dev/null
grep step /tmp/st.log
f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then grep -E "finish|adam|pack_kernel" "$f" | cut -c1-120; fi
cd $R
for rep in 1 2; do timeout 100 python tools/step_time.py 2>&1 | grep step; done
