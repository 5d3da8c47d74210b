#!/usr/bin/env python3
"""A few mixed-precision training steps of SinNeRFSystem (4096 rays, 64+64) for a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d out -o tr -- python tools/train_trace.py [graph]
   python tools/train_trace.py --summarize out/tr_kernel_trace.csv      -> per-kernel time of the LAST step, gaps, wall"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    import collections
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the fused Adam launch
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    a, b = idx[-2] + 1, idx[-1] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    per = collections.OrderedDict()
    for r in step:
        k = r["Kernel_Name"].split("(")[0][-60:]
        d = per.setdefault(k, [0, 0])
        d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("last step: %d launches, wall %.3f ms, kernels busy %.3f ms, gaps %.3f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    for k, (n, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-62s x%-3d %8.3f ms" % (k, n, ns / 1e6))
    sys.exit(0)

import torch
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
from sinnerf_amd.system import SinNeRFSystem           # noqa: E402

dev = torch.device("cuda:0")
graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
torch.manual_seed(0)
sysm = SinNeRFSystem(N_importance=64, compute_dtype="bf16", perturb=1.0, noise_std=1.0, white_back=True).to(dev)
rays = torch.from_numpy(O.lego_rays(400, 400, seed=100)[::39][:4096]).to(dev)
batch = {"rays": rays, "rgbs": torch.rand((rays.shape[0], 3), device=dev)}
for _ in range(6):
    sysm.train_step(batch, graph=graph)
torch.cuda.synchronize()
