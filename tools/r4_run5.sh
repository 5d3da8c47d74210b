#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x3prof -o x3 -- python $R/tools/x3_step_time.py bf16x3 fp32 > $R/gpurun_out/x3_step.log 2>&1
cd $R
cat gpurun_out/x3_step.log | tail -3
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/x3prof/**/x3_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-70s calls %4s  avg %9.1f us  max %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
