#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "emb" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_grads_gpu.py tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
cat > /tmp/steptime.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
from oracle import oracle_np as O
from sinnerf_amd.system import SinNeRFSystem
dev = torch.device("cuda:0")
torch.manual_seed(0)
sysm = SinNeRFSystem(N_importance=64, compute_dtype="bf16", perturb=1.0, noise_std=1.0, white_back=True).to(dev)
sysm.setup_distributed()
rays = torch.from_numpy(O.lego_rays(400, 400, seed=100)[::39][:4096]).to(dev)
batch = {"rays": rays, "rgbs": torch.rand((4096, 3), device=dev)}
for graph in (False, True):
    for _ in range(5): sysm.train_step(batch, graph=graph)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): sysm.train_step(batch, graph=graph)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%s step %.4f ms" % ("graph" if graph else "eager", best), flush=True)
PY
export R
for rep in 1 2; do
  echo -n "emb bf16: "; timeout 200 python /tmp/steptime.py 2>&1 | grep step | tr '\n' ' '; echo
  echo -n "emb fp32: "; SINNERF_EMB_FP32=1 timeout 200 python /tmp/steptime.py 2>&1 | grep step | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/emb16_ab.log
