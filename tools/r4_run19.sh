#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_r19.log 2>&1; echo "exit $?"; tail -4 gpurun_out/pytest_r19.log
python tools/x3_step_time.py bf16x3 2>&1 | tail -1
python tools/dw_x3_time.py 2>&1 | tail -9
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r19 -o x3 -- python $R/tools/x3_step_time.py bf16x3 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r19/**/*_kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 1.5: print("   %-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
