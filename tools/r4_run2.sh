#!/bin/bash
# round 4, GPU call 2: the bf16x3 kernel (fp32-level accuracy on the bf16 MFMA) -- parity at the fp32 bars, then its rate
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_bf16x3.log 2>&1; echo "exit $?"; grep -E "bf16x3|worst err|passed|failed|Error|error" gpurun_out/pytest_bf16x3.log | head -40
echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench exit $?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench2.log") if l.startswith("{")][-1])
print(len(json.dumps(d)), d.get("dropped"))
for k in ("bf16", "bf16x3", "config5_bf16", "config5_bf16x3", "config5_fp32"):
    print(k, json.dumps(d["records"].get(k)))
print("fp32", d["value"], d["roofline"]["frac"])
PY
