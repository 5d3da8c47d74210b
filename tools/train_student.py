#!/usr/bin/env python3
"""Train the 2 000-step fp32 student of tools/convergence.py ONCE on the MI355X and write its two state dicts
(VERDICT r5 "Next round" #1: parity evidence on TRAINED weights).

    python tools/train_student.py [--steps 2000] [--seed 0] [--out gpurun_out/trained_student.npz]

The npz holds `coarse.<key>` / `fine.<key>` = the reference's `NeRF.state_dict()` keys (models/nerf.py:46-103) in fp32, plus the
training recipe.  It is committed as tests/golden/trained_student.npz; oracle/gen_golden.py --trained renders it with the
UNMODIFIED reference on the CPU (eval 64+64 lego, 64+128 llff, a perturb=1 / noise_std=1 training render, autograd gradients).
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import convergence as C            # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "trained_student.npz"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sc = C.scene(dev)
    keep = {}
    run = C.run_amd("fp32", a.seed, a.steps, sc, dev, keep_state=keep)
    out = {f"{tag}.{k}": v.astype(np.float32) for tag, sd in keep.items() for k, v in sd.items()}
    out["steps"] = np.int64(a.steps)
    out["seed"] = np.int64(a.seed)
    out["final_psnr"] = np.float64(run["final_psnr"])
    out["psnr_curve"] = np.asarray(run["psnr_curve"], np.float64)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **out)
    print("trained student:", a.steps, "steps, held-out PSNR", run["final_psnr"], "->", a.out,
          os.path.getsize(a.out), "bytes")
    for tag, sd in keep.items():
        for k, v in sd.items():
            print(f"  {tag}.{k:28s} |w|max {np.abs(v).max():8.3f}  rms {np.sqrt((v.astype(np.float64) ** 2).mean()):.4f}")
