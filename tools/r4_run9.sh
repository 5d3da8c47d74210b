#!/bin/bash
# PMC of the bf16x3 training step kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3pmc -o p1 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- python $R/tools/x3_step_time.py bf16x3 > $R/gpurun_out/x3pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3pmc -o p2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -- python $R/tools/x3_step_time.py bf16x3 > $R/gpurun_out/x3pmc2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, statistics
for tag in ("p1", "p2"):
    f = glob.glob("gpurun_out/x3pmc/**/%s_counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"][:44], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 0.5: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        med = lambda c: statistics.median(per[k][c] for k in ks)
        ms = statistics.median(dur[k] for k in ks)
        if tag == "p1":
            cyc = med("GRBM_GUI_ACTIVE") / 8
            print("%-44s %7.3f ms  %8.2f Mcyc  clock %.2f GHz  mfma_busy %.3f  parked %.3f  issue-wait %.3f  active %.3f  lds-wait %.3f" % (
                name, ms, cyc / 1e6, cyc / ms / 1e6, med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, med("SQ_WAIT_ANY") / med("SQ_WAVE_CYCLES"),
                med("SQ_WAIT_INST_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_ACTIVE_INST_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_WAIT_INST_LDS") / med("SQ_WAVE_CYCLES")))
        else:
            m = max(med("SQ_INSTS_MFMA"), 1)
            print("%-44s %7.3f ms  MFMA %.3g  VALU/MFMA %.2f  LDS/MFMA %.2f  SALU/MFMA %.2f  VMEM/MFMA %.2f  bank-conflict frac %.3f  waves %d" % (
                name, ms, m, med("SQ_INSTS_VALU") / m, med("SQ_INSTS_LDS") / m, med("SQ_INSTS_SALU") / m, med("SQ_INSTS_VMEM") / m,
                med("SQ_LDS_BANK_CONFLICT") / max(med("SQ_LDS_IDX_ACTIVE"), 1), med("SQ_WAVES")))
PY
