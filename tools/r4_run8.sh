#!/bin/bash
# round 4, GPU call 8: x3 step profile after the dW reorder, then the whole GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/r4_run5.sh 2>&1 | grep -E "ms / step|dw_|mlp_" 
echo "== pytest (all)"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/convergence.json"))["summary"]
print(json.dumps(d))
PY
