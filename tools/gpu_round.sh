#!/bin/bash
# One gpurun call: tests + bench (fp32, bf16) + training bench + rocprof (kernel trace, then PMC passes).
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
# the kernel sources these profiles are measured ON, hashed here -- on the measuring box, at measurement time: the summaries
# carry this value; nothing re-stamps a profile afterwards (a hash-definition change means re-measuring, or `traffic: null`)
python -c "import bench; print(bench.kernel_sources_sha())" > gpurun_out/kernel_sources_sha.txt 2>/dev/null
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench fp32"; timeout 600 python bench.py --steps 3 --warmup 1 --full-json gpurun_out/bench_full_fp32_main.json > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-300
echo "== bench bf16"; timeout 600 python bench.py --steps 3 --warmup 1 --dtype bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench_bf16.log | cut -c1-200
echo "== train bench"; timeout 600 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1; echo "train exit $?"; tail -1 gpurun_out/train_bench.log
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof"
echo "== rocprof stats"; timeout 600 $P --stats -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-pmc > $R/gpurun_out/prof_stats.log 2>&1; echo "exit $?"
echo "== rocprof stats bf16"; timeout 600 $P --stats -o stats_bf16 -- python $R/bench.py --steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --no-extra --no-pmc > $R/gpurun_out/prof_stats_bf16.log 2>&1; echo "exit $?"
echo "== rocprof stats train"; timeout 600 $P --stats -o stats_train -- python $R/tools/train_bench.py > $R/gpurun_out/prof_stats_train.log 2>&1; echo "exit $?"
echo "== pmc1"; timeout 600 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -o pmc1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pmc > $R/gpurun_out/prof_pmc1.log 2>&1; echo "exit $?"
echo "== pmc2"; timeout 600 $P --pmc FETCH_SIZE -o pmc2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pmc > $R/gpurun_out/prof_pmc2.log 2>&1; echo "exit $?"
echo "== pmc3"; timeout 600 $P --pmc WRITE_SIZE -o pmc3 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pmc > $R/gpurun_out/prof_pmc3.log 2>&1; echo "exit $?"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"
for dt in fp32 bf16; do
  echo "== instruction mix $dt"; timeout 300 $P --pmc $C -o mix_$dt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pmc --dtype $dt > $R/gpurun_out/mix_$dt.log 2>&1; echo "exit $?"
done
echo "== pmc bf16"; timeout 600 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -o pmc1_bf16 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pmc --dtype bf16 > $R/gpurun_out/prof_pmc1_bf16.log 2>&1; echo "exit $?"
cd $R
cd /tmp
echo "== pmc train (HBM bytes of the training kernels)"
timeout 600 $P --pmc FETCH_SIZE -o pmc_train_fetch -- python $R/tools/train_bench.py > $R/gpurun_out/prof_pmc_train_fetch.log 2>&1; echo "exit $?"
timeout 600 $P --pmc WRITE_SIZE -o pmc_train_write -- python $R/tools/train_bench.py > $R/gpurun_out/prof_pmc_train_write.log 2>&1; echo "exit $?"
echo "== bf16x3 inference launch: stats, then cycles / MFMA-busy / clock"
timeout 300 $P --stats -o stats_x3 -- python $R/tools/x3_infer_time.py fp32 bf16 bf16x3 fp16 > $R/gpurun_out/x3_infer.log 2>&1; echo "exit $?"; cat $R/gpurun_out/x3_infer.log | tail -3
timeout 300 $P --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -o pmc_x3 -- python $R/tools/x3_infer_time.py fp32 bf16 bf16x3 fp16 > /dev/null 2>&1; echo "exit $?"
cd $R
python - <<'PY' | tee gpurun_out/x3_infer_pmc.txt
import csv, glob, collections, statistics
for f in glob.glob("gpurun_out/prof/**/pmc_x3_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44], r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]
        if max(dur[k] for k in ks) < 5: continue
        med = lambda c: statistics.median(per[k][c] for k in ks); ms = statistics.median(dur[k] for k in ks); cyc = med("GRBM_GUI_ACTIVE") / 8
        print("%-44s %8.3f ms %8.2f Mcyc  clock %.2f GHz  mfma_busy %.3f  parked %.3f  issue-wait %.3f" % (name, ms, cyc / 1e6, cyc / ms / 1e6,
              med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, med("SQ_WAIT_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_WAIT_INST_ANY") / med("SQ_WAVE_CYCLES")))
PY
echo "== pmc train (cycles, MFMA-busy, clock of the training kernels)"
bash tools/train_pmc.sh > gpurun_out/train_pmc.log 2>&1; echo "exit $?"; tail -12 gpurun_out/train_pmc.txt
echo "== bf16x3 training stages: generated streams vs compiler-scheduled (time, bit identity)"
python tools/x3_stage_time.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x3_stage_time.txt
echo "== fp32 inference: round-6 kernel vs the LDS-ring kernel (time, bit identity), then PMC passes"
python tools/f32_infer_ab.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/f32_infer_ab.txt
bash tools/f32_pmc.sh final > /dev/null 2>&1; echo "exit $?"; grep -A3 "f32g_kernel<false, 0, false>" gpurun_out/f32_pmc_final.txt | head -8

