#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st2 -o eager -- python $R/tools/step_trace.py eager > $R/gpurun_out/st2_eager.log 2>&1 < /dev/null; echo "exit $?"
cd $R
python - <<'PY' | tee gpurun_out/step_order.txt
import csv, glob
fs = glob.glob("gpurun_out/st2/**/eager_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    print("%9.1f us  %7.1f us  grid %-9s wg %-5s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:110]))
print("span %.3f ms" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e6))
PY
