mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_grads_gpu.py tests/test_classic_heads_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_r5e.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/pytest_r5e.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/pytest_r5e.log | head
{
for rep in 1 2 3; do
for lib in "" "$R/build/variants/lib_oldstage.so"; do
  echo "== lib=${lib:-in-tree (two accumulator chains)}"
  SINNERF_HIP_LIB=$lib python tools/chain_time.py 2>&1 | grep -v amdgpu.ids
  SINNERF_HIP_LIB=$lib python tools/fwd_train_time.py 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/ab_f32_two_chains.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | cut -c1-900 | tee gpurun_out/bench_two_chains.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/f32pmc -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -- python $R/bench.py --steps 1 --warmup 0 --no-extra --no-cpu-baseline --no-pmc > /dev/null 2>&1; echo "pmc exit $?"
cd $R
python - <<'PY' | tee gpurun_out/f32_two_chains_pmc.txt
import csv, glob, collections, statistics
for f in glob.glob("gpurun_out/f32pmc/**/p1_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:48], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in sorted(dur, key=lambda k: -dur[k])[:2]:
        cyc = per[k]["GRBM_GUI_ACTIVE"] / 8; ms = dur[k]
        print("%-48s %8.3f ms %8.2f Mcyc  clock %.2f GHz  mfma_busy %.3f  parked %.3f  issue-wait %.3f" % (k[0], ms, cyc / 1e6, cyc / ms / 1e6,
              per[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, per[k]["SQ_WAIT_ANY"] / per[k]["SQ_WAVE_CYCLES"], per[k]["SQ_WAIT_INST_ANY"] / per[k]["SQ_WAVE_CYCLES"]))
PY
