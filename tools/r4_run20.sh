#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
python tools/x3_step_time.py bf16x3 2>&1 | tail -1
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r20"
timeout 300 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -o p1 -- python $R/tools/x3_step_time.py bf16x3 > /dev/null 2>&1
timeout 300 $P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -o p2 -- python $R/tools/x3_step_time.py bf16x3 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, statistics
for tag in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/r20/**/{tag}_counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:40], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 0.5: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        med = lambda f: statistics.median(f(k) for k in ks)
        cyc = med(lambda k: per[k]["GRBM_GUI_ACTIVE"] / 8)
        if tag == "p1":
            print("%-40s %.3f ms clock %.2f busy %.3f parked %.3f issue-wait %.3f" % (name, med(lambda k: dur[k]), med(lambda k: per[k]["GRBM_GUI_ACTIVE"] / 8 / (dur[k] * 1e6)),
                  med(lambda k: per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8)), med(lambda k: per[k]["SQ_WAIT_ANY"] / per[k]["SQ_WAVE_CYCLES"]), med(lambda k: per[k]["SQ_WAIT_INST_ANY"] / per[k]["SQ_WAVE_CYCLES"])))
        else:
            m = lambda k: max(per[k]["SQ_INSTS_MFMA"], 1)
            print("%-40s lds/mfma %.2f valu/mfma %.2f salu/mfma %.2f bank-conflict %.3f" % (name, med(lambda k: per[k]["SQ_INSTS_LDS"] / m(k)), med(lambda k: (per[k]["SQ_INSTS_VALU"] - per[k]["SQ_INSTS_MFMA"]) / m(k)),
                  med(lambda k: per[k]["SQ_INSTS_SALU"] / m(k)), med(lambda k: per[k]["SQ_LDS_BANK_CONFLICT"] / max(per[k]["SQ_LDS_IDX_ACTIVE"], 1))))
PY
