#!/usr/bin/env python3
"""Time the fused bf16 MLP forward launch (fine-pass shape of the bench: 160 000 rays x 128 samples) for one or more builds
of the library.  usage: mlp_time.py [lib.so ...]   (each lib is timed in a fresh subprocess via SINNERF_HIP_LIB)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import numpy as np
    import torch
    from oracle import oracle_np as O
    import sinnerf_amd
    from sinnerf_amd import rendering
    dev = torch.device("cuda:0")
    flags = int(sys.argv[2])
    p = O.init_params(1, True)
    m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    m = m.to(dev).eval()
    rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
    z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
    with torch.no_grad():
        for _ in range(2):
            out = rendering._mlp(m, rays, z, False, flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = rendering._mlp(m, rays, z, False, flags)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    pts = rays.shape[0] * 128
    print(json.dumps({"ms": ms, "tflops": 1186816 * pts / ms / 1e9, "frac": 1186816 * pts / ms / 1e9 / 2500,
                      "cyc_per_mfma_at_2.4GHz": ms * 1e-3 * 2.4e9 / (pts / 64 / 1024 * 2320), "finite": bool(torch.isfinite(out).all())}))
    sys.exit(0)

libs = sys.argv[1:] or [os.path.join(REPO, "sinnerf_amd", "csrc", "libsinnerf_hip.so")]
for lib in libs:
    flags = "0"
    if lib.endswith(":legacy"):
        lib, flags = lib[:-7], "2"
    env = dict(os.environ, SINNERF_HIP_LIB=lib)
    r = subprocess.run([sys.executable, __file__, "--child", flags], env=env, capture_output=True, text=True, timeout=300)
    print("%-40s %s" % (os.path.basename(lib) + (":legacy" if flags == "2" else ""), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]))
