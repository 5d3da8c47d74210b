#!/bin/bash
# PMC passes over the training kernels of tools/train_bench.py (fine-pass dispatch of each kernel): clock, MFMA-busy, waits,
# LDS conflicts.  usage: tools/train_pmc.sh   (summary: gpurun_out/train_pmc.txt)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tpmc"
timeout 300 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -o p1 -- python $R/tools/train_bench.py > $R/gpurun_out/tpmc1.log 2>&1; echo "p1 exit $?"
timeout 300 $P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -o p2 -- python $R/tools/train_bench.py > $R/gpurun_out/tpmc2.log 2>&1; echo "p2 exit $?"
cd $R
python - <<'PY' | tee gpurun_out/train_pmc.txt
import csv, collections, glob, json, statistics
out = {"command": "rocprofv3 --kernel-trace --pmc <SQ counters> -- python tools/train_bench.py (two passes)",
       "note": "median over the fine-pass-sized dispatches of each kernel (the first dispatches after an allocation run at lower clocks)", "kernels": {}}
for tag in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/tpmc/**/{tag}_counter_collection.csv", recursive=True)
    if not fs: print(tag, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:60], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 0.25: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        med = lambda f: statistics.median(f(k) for k in ks)
        cyc = med(lambda k: per[k]["GRBM_GUI_ACTIVE"] / 8)
        rec = out["kernels"].setdefault(name, {})
        rec.update(dispatches=len(ks), ms=med(lambda k: dur[k]), gpu_cycles=cyc, clock_ghz=med(lambda k: per[k]["GRBM_GUI_ACTIVE"] / 8 / (dur[k] * 1e6)))
        if tag == "p1":
            rec.update(mfma_busy=med(lambda k: per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8)),
                       wait_any=med(lambda k: per[k]["SQ_WAIT_ANY"] / per[k]["SQ_WAVE_CYCLES"]))
        else:
            m = lambda k: max(per[k]["SQ_INSTS_MFMA"], 1)
            rec.update(lds_per_mfma=med(lambda k: per[k]["SQ_INSTS_LDS"] / m(k)), valu_per_mfma=med(lambda k: (per[k]["SQ_INSTS_VALU"] - per[k]["SQ_INSTS_MFMA"]) / m(k)),
                       salu_per_mfma=med(lambda k: per[k]["SQ_INSTS_SALU"] / m(k)),
                       lds_bank_conflict_frac=med(lambda k: per[k]["SQ_LDS_BANK_CONFLICT"] / max(per[k]["SQ_LDS_IDX_ACTIVE"], 1)))
for name, rec in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
    print("%-60s" % name, " ".join("%s=%.3g" % kv for kv in rec.items()))
json.dump(out, open("gpurun_out/train_pmc_cycles.json", "w"), indent=1)
PY
