#!/usr/bin/env python3
"""Convergence-length PSNR parity of the training paths (VERDICT r3 "Next round" #4, SURVEY §8d "PSNR parity protocol").

Teacher scene: the fp32 render of the teacher weights (oracle_np.init_params(seed, teacher=True)) of a 400x400 lego-shaped frame
is the ground truth; a student (different init seeds) is trained on random 1024-ray batches of it with the TRAINING DEFAULTS of the
reference (perturb=1, noise_std=1: opt.py:25-28; MSE coarse + fine: losses.py:12-22; Adam lr 5e-4 eps 1e-8:
utils/__init__.py:19-21) and scored on held-out rays of a second camera pose (PSNR of the deterministic fine render,
metrics.py:14-15, as validation_step does: sinnerf.py:556-577).  Three paths from the same initial weights, the same batches and
the same device-RNG seed (all three consume the generator in the reference's order):

  bf16   sinnerf_amd, mixed precision (bf16-operand forward / chain / weight gradients over a bf16 training state)
  bf16x3 sinnerf_amd, fp32-level accuracy on the bf16 MFMA (3-term hi/lo splits), training state stored as the (hi, lo) pairs
  fp32   sinnerf_amd, fp32 MFMA kernels
  ref    the UNMODIFIED reference render_rays + NeRF modules (oracle/_ref) as PyTorch-ROCm eager ops on the same GPU

usage: python tools/convergence.py [--steps 2000] [--seeds 0 1 2] [--ref-seeds 0] [--out gpurun_out/convergence.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_np as O          # noqa: E402

BATCH = 1024
EVAL_EVERY = 100


def scene(dev):
    import sinnerf_amd
    teacher = []
    for s in (0, 1):
        m = sinnerf_amd.NeRF(use_new_activation=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(s, teacher=True).items()})
        teacher.append(m.to(dev).eval())
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    train_rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)).to(dev)
    held = torch.from_numpy(np.ascontiguousarray(O.lego_rays(400, 400, seed=1)[::37][:4096])).to(dev)
    with torch.no_grad():
        tgt = sinnerf_amd.render_rays(teacher, emb, train_rays, 64, False, 0, 0, 64, 32768, True)["rgb_fine"]
        tgt_held = sinnerf_amd.render_rays(teacher, emb, held, 64, False, 0, 0, 64, 32768, True)["rgb_fine"]
    return train_rays, tgt, held, tgt_held


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a - b) ** 2)))


def batches(n_rays, steps, seed):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    return [torch.randint(0, n_rays, (BATCH,), generator=g) for _ in range(steps)]


def run_amd(dtype, seed, steps, sc, dev, keep_state=None):
    from sinnerf_amd.system import SinNeRFSystem
    train_rays, tgt, held, tgt_held = sc
    sysm = SinNeRFSystem(N_importance=64, compute_dtype=dtype, perturb=1.0, noise_std=1.0, white_back=True, lr=5e-4,
                         decay_step=[10 ** 9])
    for m, s in zip(sysm.models, (10 + 2 * seed, 11 + 2 * seed)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(s, teacher=True).items()})
    sysm = sysm.to(dev)
    sysm.configure_optimizers()
    idx = batches(train_rays.shape[0], steps, seed)
    curve, losses = [], []

    def evaluate(step):
        hp = sysm.hparams
        keep = (hp.perturb, hp.noise_std)
        hp.perturb, hp.noise_std = 0, 0
        state = torch.cuda.get_rng_state(dev)                      # the evaluation must not move the training RNG stream
        with torch.no_grad():
            r = sysm(held)["rgb_fine"]
        torch.cuda.set_rng_state(state, dev)
        hp.perturb, hp.noise_std = keep
        curve.append((step, psnr(r, tgt_held)))

    torch.manual_seed(seed)
    t0 = time.perf_counter()
    for i in range(steps):
        if i % EVAL_EVERY == 0:
            evaluate(i)
        ii = idx[i].to(dev)
        out = sysm.train_step({"rays": train_rays[ii], "rgbs": tgt[ii]})
        if i % 20 == 0:
            losses.append((i, float(out["loss"])))
    evaluate(steps)
    torch.cuda.synchronize()
    if keep_state is not None:                                     # tools/train_student.py: the trained weights as a fixture
        for tag, m in zip(("coarse", "fine"), sysm.models):
            keep_state[tag] = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    return {"path": dtype, "seed": seed, "psnr_curve": curve, "loss_curve": losses, "final_psnr": curve[-1][1],
            "seconds": time.perf_counter() - t0}


def run_ref(seed, steps, sc, dev):
    from oracle import stage_ref
    rendering, nerf = stage_ref.load()
    train_rays, tgt, held, tgt_held = sc
    models = []
    for s in (10 + 2 * seed, 11 + 2 * seed):
        m = nerf.NeRF(use_new_activation=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(s, teacher=True).items()})
        models.append(m.to(dev).train())
    emb = [nerf.Embedding(3, 10), nerf.Embedding(3, 4)]
    opt = torch.optim.Adam([p for m in models for p in m.parameters()], lr=5e-4, eps=1e-8)
    idx = batches(train_rays.shape[0], steps, seed)
    curve, losses = [], []

    def evaluate(step):
        state = torch.cuda.get_rng_state(dev)
        with torch.no_grad():
            r = torch.cat([rendering.render_rays(models, emb, held[j:j + 1024], 64, False, 0, 0, 64, 32768, True)["rgb_fine"]
                           for j in range(0, held.shape[0], 1024)], 0)
        torch.cuda.set_rng_state(state, dev)
        curve.append((step, psnr(r, tgt_held)))

    torch.manual_seed(seed)
    t0 = time.perf_counter()
    for i in range(steps):
        if i % EVAL_EVERY == 0:
            evaluate(i)
        ii = idx[i].to(dev)
        opt.zero_grad(set_to_none=True)
        r = rendering.render_rays(models, emb, train_rays[ii], 64, False, 1.0, 1.0, 64, 32768, True)
        loss = torch.mean((r["rgb_coarse"] - tgt[ii]) ** 2) + torch.mean((r["rgb_fine"] - tgt[ii]) ** 2)     # losses.py:12-22
        loss.backward()
        opt.step()
        if i % 20 == 0:
            losses.append((i, float(loss)))
    evaluate(steps)
    torch.cuda.synchronize()
    return {"path": "ref", "seed": seed, "psnr_curve": curve, "loss_curve": losses, "final_psnr": curve[-1][1],
            "seconds": time.perf_counter() - t0}


def run_all(steps=2000, seeds=(0, 1, 2), ref_seeds=(0,), dev=None):
    dev = dev or torch.device("cuda:0")
    sc = scene(dev)
    runs = []
    for s in seeds:
        for dt in ("bf16", "bf16x3", "fp32"):
            runs.append(run_amd(dt, s, steps, sc, dev))
    from oracle import stage_ref
    if stage_ref.available():
        for s in ref_seeds:
            runs.append(run_ref(s, steps, sc, dev))
    fin = lambda path: [r["final_psnr"] for r in runs if r["path"] == path]
    summ = {"steps": steps, "batch_rays": BATCH, "seeds": list(seeds), "ref_seeds": list(ref_seeds) if fin("ref") else [],
            "final_psnr": {p: fin(p) for p in ("bf16", "bf16x3", "fp32", "ref") if fin(p)},
            "mean_final_psnr": {p: float(np.mean(fin(p))) for p in ("bf16", "bf16x3", "fp32", "ref") if fin(p)},
            "seconds": {p: [round(r["seconds"], 1) for r in runs if r["path"] == p] for p in ("bf16", "bf16x3", "fp32", "ref")}}
    summ["bf16_minus_fp32_dB"] = summ["mean_final_psnr"]["bf16"] - summ["mean_final_psnr"]["fp32"]
    summ["bf16x3_minus_fp32_dB"] = summ["mean_final_psnr"]["bf16x3"] - summ["mean_final_psnr"]["fp32"]
    return {"protocol": __doc__.split("usage:")[0].strip(), "summary": summ, "runs": runs}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--ref-seeds", type=int, nargs="*", default=[0])
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "convergence.json"))
    a = ap.parse_args()
    res = run_all(a.steps, a.seeds, a.ref_seeds)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res["summary"]))
