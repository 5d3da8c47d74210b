#!/usr/bin/env python3
"""Fine-pass MLP launch (160 000 rays x 128 samples = 20.48 M points) of the inference kernels, a few repetitions per compute_dtype:
run under rocprofv3 (--kernel-trace --stats, or --pmc) for per-kernel time, cycles, MFMA-busy.  usage: x3_infer_time.py [fp32 bf16 bf16x3 ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import rendering                      # noqa: E402

dev = torch.device("cuda:0")
rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
for dt in (sys.argv[1:] or ["bf16x3"]):
    m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=dt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(2):
            rendering._mlp(m, rays, z, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            rendering._mlp(m, rays, z, False)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-7s fine pass %.3f ms  %.1f TFLOP/s algorithmic" % (dt, ms, 1186816 * rays.shape[0] * 128 / ms / 1e9))
