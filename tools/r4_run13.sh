#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_bf16x3.log 2>&1; echo "exit $?"; tail -3 gpurun_out/pytest_bf16x3.log
bash tools/r4_run9.sh 2>&1 | tail -6
python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import rendering
dev = torch.device("cuda:0")
rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
for dt in ("bf16x3", "bf16", "fp32"):
    m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=dt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(2): rendering._mlp(m, rays, z, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): rendering._mlp(m, rays, z, False)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-7s fine-pass MLP launch %.3f ms  %.1f TF algorithmic" % (dt, ms, 1186816 * rays.shape[0] * 128 / ms / 1e9))
PY
