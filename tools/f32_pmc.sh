#!/bin/bash
# PMC passes over tools/f32_infer_ab.py (the round-6 fp32 inference kernel beside the LDS-ring one): where do the non-MFMA cycles go?
# usage (on the GPU box): bash tools/f32_pmc.sh [tag]    -> gpurun_out/f32_pmc_<tag>.txt
R=$PWD; TAG=${1:-run}; export TMPDIR=/tmp; mkdir -p gpurun_out; cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_f32"
rm -rf $R/gpurun_out/prof_f32
timeout 300 $P --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -o p1 -- python $R/tools/f32_infer_ab.py 1 > /dev/null 2>&1; echo "pass 1 exit $?"
timeout 300 $P --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS -o p2 -- python $R/tools/f32_infer_ab.py 1 > /dev/null 2>&1; echo "pass 2 exit $?"
timeout 300 $P --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM -o p3 -- python $R/tools/f32_infer_ab.py 1 > /dev/null 2>&1; echo "pass 3 exit $?"
timeout 300 $P --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum -o p4 -- python $R/tools/f32_infer_ab.py 1 > /dev/null 2>&1; echo "pass 4 exit $?"
cd $R
python - "$TAG" <<'PY' | tee gpurun_out/f32_pmc_$1.txt
import csv, glob, collections, statistics, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_f32/**/*_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, d in dur.items():
        if d < 100: continue                      # the full (non sigma-only) fine-pass launches
        for c, v in per[k].items(): rows[k[0]][c].append(v)
        rows[k[0]]["ms"].append(d)
for name, c in sorted(rows.items()):
    m = {k: statistics.median(v) for k, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(name)
    print("   %.3f ms, %.1f Mcyc/XCD, clock %.2f GHz" % (m["ms"], cyc / 1e6, cyc / m["ms"] / 1e6 if m["ms"] else 0))
    wc = m.get("SQ_WAVE_CYCLES")
    for k in sorted(m):
        if k in ("ms", "GRBM_GUI_ACTIVE"): continue
        extra = ""
        if wc and k.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_BUSY")): extra = "  = %.4f of SQ_WAVE_CYCLES" % (m[k] / wc)
        if k == "SQ_VALU_MFMA_BUSY_CYCLES": extra = "  = %.4f busy (/1024 waves-slots /cycles)" % (m[k] / 1024 / cyc)
        print("   %-34s %16.0f%s" % (k, m[k], extra))
PY
