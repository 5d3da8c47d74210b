#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
python tools/x3_step_time.py bf16x3 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r21 -o x3 -- python $R/tools/x3_step_time.py bf16x3 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, statistics
for f in glob.glob("gpurun_out/r21/**/x3_kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 2000: continue
        mx = max(v); big = [x for x in v if x > 0.75 * mx]; small = [x for x in v if x <= 0.75 * mx]
        print("%-50s n=%3d fine %.1f us coarse %.1f us" % (k, len(v), statistics.median(big), statistics.median(small) if small else 0))
PY
