#!/usr/bin/env python3
"""One training step (4096 rays, 64+64, perturb=1, noise_std=1) per compute_dtype given on the command line, a few repetitions:
run under `rocprofv3 --kernel-trace --stats` for the per-kernel times.  usage: x3_step_time.py [fp32 bf16x3 ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
import sinnerf_amd                                     # noqa: E402

dev = torch.device("cuda:0")
N = 4096
rays = torch.from_numpy(O.lego_rays(400, 400, 0)[:: 160000 // N][:N]).to(dev)
tgt = torch.rand((N, 3), device=dev)
emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
for dt in (sys.argv[1:] or ["fp32", "bf16x3"]):
    models = []
    for seed in (0, 1):
        m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=dt)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(seed, True).items()})
        models.append(m.to(dev).train())

    def step():
        for m in models:
            m.zero_grad(set_to_none=True)
        res = sinnerf_amd.render_rays(models, emb, rays, 64, False, 1.0, 1.0, 64, 32768, True)
        (((res["rgb_fine"] - tgt) ** 2).mean() + ((res["rgb_coarse"] - tgt) ** 2).mean()).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("%-8s %.3f ms / step" % (dt, (time.perf_counter() - t0) / 5 * 1e3))
