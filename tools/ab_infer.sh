#!/bin/bash
# A/B of bf16 INFERENCE kernel builds in GPU cycles, MFMA-busy and effective clock: one PMC pass per build over tools/mlp_time.py
# (fine-pass shape: 160 000 rays x 128 samples).  usage: tools/ab_infer.sh tag [variant.so ...]   (in-tree library first)
export TMPDIR=/tmp
R=$PWD
tag=$1; shift
mkdir -p gpurun_out
cd /tmp
i=0
for lib in "" "$@"; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abi_$tag -o l${i} --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -- python $R/tools/mlp_time.py ${lib:+$R/$lib} > $R/gpurun_out/abi_${tag}_l${i}.log 2>&1
  i=$((i+1))
done
cd $R
python - "$tag" "in-tree" "$@" <<'PY' | tee gpurun_out/ab_infer_$1.txt
import csv, sys, glob, collections, statistics
tag = sys.argv[1]
print("%-28s %-30s %9s %10s %9s %9s %8s" % ("build", "kernel", "ms", "Mcycles", "clock GHz", "mfma_busy", "parked"))
for i, lib in enumerate(sys.argv[2:]):
    fs = glob.glob(f"gpurun_out/abi_{tag}/**/l{i}_counter_collection.csv", recursive=True)
    if not fs:
        print(lib, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:30], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 2.0: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        cyc = statistics.median(per[k]["GRBM_GUI_ACTIVE"] / 8 for k in ks)
        ms = statistics.median(dur[k] for k in ks)
        busy = statistics.median(per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8) for k in ks)
        parked = statistics.median(per[k]["SQ_WAIT_ANY"] / max(per[k]["SQ_WAVE_CYCLES"], 1) for k in ks)
        print("%-28s %-30s %9.3f %10.3f %9.3f %9.3f %8.3f" % (lib[-28:], name, ms, cyc / 1e6, cyc / ms / 1e6, busy, parked))
PY
