#!/bin/bash
# A/B of builds of the bf16 inference kernel: wall time (tools/mlp_time.py) and GPU cycles / MFMA-busy from one PMC pass each
# usage: tools/gpu_ab.sh lib1.so lib2.so ...
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
python tools/mlp_time.py "$@" "$1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_time.log
cd /tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  SINNERF_HIP_LIB=$R/$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab -o $name --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -- python $R/tools/mlp_time.py --child 0 > $R/gpurun_out/ab_$name.log 2>&1
done
cd $R
python - "$@" <<'PY'
import csv, sys, os, collections
for lib in sys.argv[1:]:
    name = os.path.basename(lib)[:-3]
    p = f"gpurun_out/ab/{name}_counter_collection.csv"
    if not os.path.exists(p):
        print(name, "no counters"); continue
    agg = collections.defaultdict(float); n = set(); dur = {}
    for r in csv.DictReader(open(p)):
        if "v3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = len(n); cyc = agg["GRBM_GUI_ACTIVE"] / k / 8; ms = sum(dur.values()) / k
    print("%-22s cycles/launch %.3fM  clock %.3f GHz  ms(profiled) %.3f  mfma busy %.4f  wait_any %.3f  valu/mfma %.2f" % (
        name, cyc / 1e6, cyc / (ms * 1e6), ms, agg["SQ_VALU_MFMA_BUSY_CYCLES"] / k / 1024 / cyc, agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"],
        (agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"]) / agg["SQ_INSTS_MFMA"]))
PY
