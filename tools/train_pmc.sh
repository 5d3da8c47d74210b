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
import csv, collections, glob
for tag in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/tpmc/**/{tag}_counter_collection.csv", recursive=True)
    if not fs: print(tag, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:60], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    best = {}
    for k, d in dur.items():
        if k[0] not in best or d > dur[best[k[0]]]: best[k[0]] = k
    for name, k in sorted(best.items(), key=lambda kv: -dur[kv[1]])[:8]:
        c = per[k]; ms = dur[k]; cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        line = "%-60s ms %.3f clock %.3f GHz" % (name, ms, cyc / (ms * 1e6) if ms else 0)
        if tag == "p1":
            line += "  mfma_busy %.3f  wait_any %.3f  wait_inst %.3f  active_inst %.3f" % (
                c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"])
        else:
            m = max(c["SQ_INSTS_MFMA"], 1)
            line += "  lds_conflict/idx_active %.3f  lds/mfma %.2f  valu/mfma %.2f  salu/mfma %.2f  wait_lds_raw %.3g" % (
                c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_INSTS_LDS"] / m,
                (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / m, c["SQ_INSTS_SALU"] / m, c["SQ_WAIT_INST_LDS"])
        print(tag, line)
PY
