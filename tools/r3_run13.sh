#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for mode in eager graph; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st -o $mode -- python $R/tools/step_trace.py $mode > $R/gpurun_out/st_$mode.log 2>&1; echo "$mode exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/step_trace.txt
import csv, glob, collections
for mode in ("eager", "graph"):
    fs = glob.glob(f"gpurun_out/st/**/{mode}_kernel_trace.csv", recursive=True)
    if not fs: print(mode, "no trace"); continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    # the last step = from the last adam_kernel but one to the last adam_kernel
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    a, b = idx[-2] + 1, idx[-1] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print("== %s: last step %d kernels, span %.3f ms, sum of kernel durations %.3f ms, idle %.3f ms" % (mode, len(step), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    agg = collections.OrderedDict()
    for r in step:
        n = r["Kernel_Name"][:70]
        d = agg.setdefault(n, [0, 0])
        d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("   %-72s x%-3d %8.1f us" % (n, c, d / 1e3))
    gaps = [(int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"])) / 1e3 for i in range(len(step) - 1)]
    print("   gaps: n=%d mean %.1f us, max %.1f us, >10us: %d" % (len(gaps), sum(gaps) / len(gaps), max(gaps), sum(g > 10 for g in gaps)))
PY
