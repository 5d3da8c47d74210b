#!/bin/bash
# round 4, GPU call 3: bf16x3 training forward (fp32 state) -- parity, golden gradients, step time; sustained-peak ubench with ReLU-like operands
mkdir -p gpurun_out
export TMPDIR=/tmp

echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_bf16x3.log 2>&1; echo "exit $?"; grep -E "bf16x3|worst err|passed|failed|Error|error|assert" gpurun_out/pytest_bf16x3.log | head -40
echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench3.log 2>&1; echo "bench exit $?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench3.log") if l.startswith("{")][-1])
print(len(json.dumps(d)), d.get("dropped"))
f = json.load(open("gpurun_out/bench_full_fp32_n1.json"))
for k in ("train_step", "train_step_bf16x3", "train_step_bf16"):
    print(k, json.dumps(f.get(k)))
for k in ("train_cfg2_fp32", "train_cfg2_bf16x3", "train_cfg2_bf16"):
    r = f["records"].get(k); print(k, r and {x: r[x] for x in ("ms_per_step", "train_rays_per_s", "frac_of_mfma_peak", "loss") if x in r} or r)
PY
