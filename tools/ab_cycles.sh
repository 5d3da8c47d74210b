#!/bin/bash
# A/B of training-kernel builds in GPU CYCLES (wall time follows the clock, which moves +-5 % with the thermal state):
# one PMC pass per build over tools/fwd_train_time.py and tools/chain_time.py; median over the fine-pass dispatches.
# usage: tools/ab_cycles.sh [variant.so ...]   (the in-tree library is always measured first; log: gpurun_out/ab_cycles.log)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
[ -z "$SKIP_TESTS" ] && python -m pytest tests/test_grads_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/ab_cycles.log
cd /tmp
i=0
for lib in "" "$@"; do
  for sc in ${SCRIPTS:-fwd_train_time chain_time}; do
    SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abc -o l${i}_$sc --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -- python $R/tools/$sc.py > $R/gpurun_out/abc_l${i}_$sc.log 2>&1
  done
  i=$((i+1))
done
cd $R
python - "main" "$@" <<'PY' | tee -a gpurun_out/ab_cycles.log
import csv, sys, glob, collections, statistics
for i, lib in enumerate(sys.argv[1:]):
    for sc in ("fwd_train_time", "chain_time"):
        fs = glob.glob(f"gpurun_out/abc/**/l{i}_{sc}_counter_collection.csv", recursive=True)
        if not fs: print(lib, sc, "no counters"); continue
        per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
        for r in csv.DictReader(open(fs[0])):
            k = (r["Kernel_Name"][:48], r["Dispatch_Id"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for name in sorted({k[0] for k in dur}):
            ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
            if mx < 2.0: continue
            ks = [k for k in ks if dur[k] > 0.6 * mx]
            cyc = statistics.median(per[k]["GRBM_GUI_ACTIVE"] / 8 for k in ks)
            busy = statistics.median(per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8) for k in ks)
            ms = statistics.median(dur[k] for k in ks)
            print("%-28s %-48s n=%d  cycles %.3fM  mfma_busy %.3f  ms %.3f" % (lib[-28:], name, len(ks), cyc / 1e6, busy, ms))
PY
