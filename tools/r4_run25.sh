#!/bin/bash
# which staging instruction class owns the LDS bank conflicts of the bf16-state training forward: timing builds of
# the generated trunk without the staging ds_write_b128s (stw0), without the ds_read_b128s (str0), without the round trip (st0)
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
cd /tmp
for lib in "" build/variants/lib_stw0.so build/variants/lib_str0.so build/variants/lib_st0.so; do
  n=$(basename ${lib:-shipped} .so)
  SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r25 -o $n --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python $R/tools/fwd_t_time.py > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r25_bank_conflicts.txt
import csv, glob, collections, statistics
print("%-14s %8s %10s %14s %14s %10s %10s" % ("build", "ms", "Mcycles", "LDS active", "bank conflict", "fraction", "LDS insts"))
for f in sorted(glob.glob("gpurun_out/r25/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        if "mlp_fwd_bf16_t" not in r["Kernel_Name"]: continue
        k = r["Dispatch_Id"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ks = list(dur)
    if not ks: continue
    med = lambda c: statistics.median(per[k][c] for k in ks)
    print("%-14s %8.3f %10.3f %14.3e %14.3e %10.3f %10.3e   addr-conflict %.3e unaligned %.3e" % (f.split("/")[-1].replace("_counter_collection.csv", ""), statistics.median(dur.values()), med("GRBM_GUI_ACTIVE") / 8e6,
          med("SQ_LDS_IDX_ACTIVE"), med("SQ_LDS_BANK_CONFLICT"), med("SQ_LDS_BANK_CONFLICT") / max(med("SQ_LDS_IDX_ACTIVE"), 1), med("SQ_INSTS_LDS"), med("SQ_LDS_ADDR_CONFLICT"), med("SQ_LDS_UNALIGNED_STALL")))
PY
