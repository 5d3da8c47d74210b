#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_r28.log 2>&1; echo "exit $?"; tail -2 gpurun_out/pytest_r28.log
for lib in build/variants/lib_head.so ""; do
  echo "== ${lib:-in-tree}"
  SINNERF_HIP_LIB=${lib:+$PWD/$lib} python tools/x3_infer_time.py bf16x3 2>&1 | tail -1
  SINNERF_HIP_LIB=${lib:+$PWD/$lib} python tools/x3_step_time.py bf16x3 2>&1 | tail -1
  cd /tmp
  SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r28 -o x3 -- python $R/tools/x3_step_time.py bf16x3 > /dev/null 2>&1
  cd $R
  python - <<'PY'
import csv, glob, collections, statistics
for f in glob.glob("gpurun_out/r28/**/x3_kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 2000: continue
        mx = max(v); big = [x for x in v if x > 0.75 * mx]; small = [x for x in v if x <= 0.75 * mx]
        print("   %-50s n=%3d fine %.1f us coarse %.1f us" % (k, len(v), statistics.median(big), statistics.median(small) if small else 0))
PY
  rm -rf gpurun_out/r28
done
