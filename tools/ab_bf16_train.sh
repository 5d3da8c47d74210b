#!/bin/bash
# A/B of bf16-state training kernels: tests, wall time per stage, and GPU cycles / MFMA-busy from one PMC pass per build.
# usage: tools/ab_bf16_train.sh [variant.so ...]   (log: gpurun_out/ab_bf16_train.log)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
{
[ -z "$SKIP_TESTS" ] && python -m pytest tests/test_grads_gpu.py tests/test_bf16_configs_gpu.py tests/test_autograd_optimizer_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for lib in "" "$@"; do echo -n "lib=${lib:-main}  "; SINNERF_HIP_LIB=$lib python tools/bf16_stage_time.py 2>&1 | grep "S="; done; done
} | tee gpurun_out/ab_bf16_train.log
cd /tmp
i=0
for lib in "" "$@"; do
  SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abb -o l$i --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -- python $R/tools/bf16_stage_time.py > $R/gpurun_out/abb_l$i.log 2>&1
  i=$((i+1))
done
cd $R
python - "main" "$@" <<'PY' | tee -a gpurun_out/ab_bf16_train.log
import csv, sys, glob, collections, statistics
for i, lib in enumerate(sys.argv[1:]):
    fs = glob.glob(f"gpurun_out/abb/**/l{i}_counter_collection.csv", recursive=True)
    if not fs: print(lib, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:48], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 0.5: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        med = lambda f: statistics.median(f(k) for k in ks)
        print("%-24s %-48s n=%d  cycles %.3fM  mfma_busy %.3f  wait_any %.3f  ms %.3f" % (
            lib[-24:], name, len(ks), med(lambda k: per[k]["GRBM_GUI_ACTIVE"] / 8) / 1e6,
            med(lambda k: per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8)),
            med(lambda k: per[k]["SQ_WAIT_ANY"] / per[k]["SQ_WAVE_CYCLES"]), med(lambda k: dur[k])))
PY
