#!/bin/bash
# round 3, GPU call 2: the hand-scheduled training forward -- bit identity vs the compiler-scheduled kernel, the round-3 tests that
# did not run in call 1, the training suites, stage timing A/B and one PMC pass
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest round3 + grads + bf16 configs"; timeout 1500 python -m pytest tests/test_round3_gpu.py tests/test_grads_gpu.py tests/test_bf16_configs_gpu.py tests/test_round2_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_b.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/pytest_b.log
echo "== stage time A/B (fine pass 524288 points: fwd_train / chain / dW)"
for rep in 1 2; do
  echo -n "hand-scheduled      "; python tools/bf16_stage_time.py 2>&1 | grep "S="
  echo -n "compiler-scheduled  "; SINNERF_COMPILER_SCHEDULED=1 python tools/bf16_stage_time.py 2>&1 | grep "S="
done | tee gpurun_out/stage_ab.log
echo "== train pmc"; bash tools/train_pmc.sh > gpurun_out/train_pmc.log 2>&1; grep "bf16" gpurun_out/train_pmc.txt
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tf"
timeout 300 $P --pmc FETCH_SIZE -o f -- python $R/tools/bf16_stage_time.py > $R/gpurun_out/tf_f.log 2>&1; echo "fetch exit $?"
timeout 300 $P --pmc WRITE_SIZE -o w -- python $R/tools/bf16_stage_time.py > $R/gpurun_out/tf_w.log 2>&1; echo "write exit $?"
cd $R
python - <<'PY'
import csv, glob, collections, statistics
res = {}
for tag, cnt in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    fs = glob.glob(f"gpurun_out/tf/**/{tag}_counter_collection.csv", recursive=True)
    if not fs: print(tag, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != cnt: continue
        k = (r["Kernel_Name"][:60], r["Dispatch_Id"]); per[k][cnt] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for name in sorted({k[0] for k in dur}):
        ks = [k for k in dur if k[0] == name]; mx = max(dur[k] for k in ks)
        if mx < 0.3: continue
        ks = [k for k in ks if dur[k] > 0.6 * mx]
        res.setdefault(name, {})[cnt] = statistics.median(per[k][cnt] for k in ks) * 1024 * (2 if cnt == "FETCH_SIZE" else 1)
        res[name]["ms"] = statistics.median(dur[k] for k in ks)
for name, v in res.items():
    b = v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)
    print("%-62s ms %.3f  bytes/point %.0f  TB/s %.2f" % (name, v["ms"], b / 524288, b / v["ms"] / 1e9))
PY
