#!/bin/bash
# round 3, GPU call 1: new tests + full suite, smoke, default bench, counter list, baseline PMC of the bf16 training kernels and
# the wait split of the bf16 inference kernel
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-400
rocprofv3 -L > gpurun_out/counters.txt 2>&1
echo "== train pmc"; bash tools/train_pmc.sh > gpurun_out/train_pmc.log 2>&1; tail -14 gpurun_out/train_pmc.txt
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w3"
timeout 300 $P --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -o w1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --dtype bf16 > $R/gpurun_out/w3_1.log 2>&1; echo "w1 exit $?"
timeout 300 $P --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -o w2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --dtype bf16 > $R/gpurun_out/w3_2.log 2>&1; echo "w2 exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("w1", "w2"):
    fs = glob.glob(f"gpurun_out/w3/**/{tag}_counter_collection.csv", recursive=True)
    if not fs: print(tag, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        if "v3" not in r["Kernel_Name"]: continue
        k = r["Dispatch_Id"]; per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if not dur: continue
    k = max(dur, key=dur.get)
    print(tag, "fine-pass v3 dispatch: ms %.3f" % dur[k], {n: "%.4g" % v for n, v in per[k].items()})
PY
