mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
: > gpurun_out/x3_variants2.txt
for lib in sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_onebar.so build/variants/lib_nosign.so build/variants/lib_norelu.so build/variants/lib_nosignrelu.so build/variants/lib_novstore.so build/variants/lib_nostage.so sinnerf_amd/csrc/libsinnerf_hip.so; do
  echo "== $lib" | tee -a gpurun_out/x3_variants2.txt
  SINNERF_HIP_LIB=$R/$lib timeout 200 python tools/x3_stage_time.py 20 2>&1 | grep -E "generated|compiler" | tee -a gpurun_out/x3_variants2.txt
done
cd /tmp
for lib in build/variants/lib_onebar.so build/variants/lib_nosignrelu.so build/variants/lib_novstore.so build/variants/lib_nostage.so; do
  n=$(basename $lib .so)
  SINNERF_HIP_LIB=$R/$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3pmc2 -o $n --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -- python $R/tools/x3_stage_time.py 3 > /dev/null 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/x3_variants2_pmc.txt
import csv, glob, collections, statistics, os
for f in sorted(glob.glob("gpurun_out/x3pmc2/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("==", os.path.basename(f))
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]
        if max(dur[k] for k in ks) < 0.5 or "fwd_bf16x3" not in name: continue
        med = lambda c: statistics.median(per[k][c] for k in ks); ms = statistics.median(dur[k] for k in ks); cyc = med("GRBM_GUI_ACTIVE") / 8
        print("%-44s %8.3f ms %8.2f Mcyc  clock %.2f GHz  mfma_busy %.3f  parked %.3f  issue-wait %.3f  active %.3f" % (name, ms, cyc / 1e6, cyc / ms / 1e6,
              med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, med("SQ_WAIT_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_WAIT_INST_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_ACTIVE_INST_ANY") / med("SQ_WAVE_CYCLES")))
PY
