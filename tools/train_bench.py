#!/usr/bin/env python3
"""Training-step timing of the hot path (fwd + bwd of render_rays on one 4096-ray batch, 64+64, fp32,
perturb=1, noise_std=1), with a per-stage breakdown from HIP events.  Writes gpurun_out/train_bench.json."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import autograd as A                  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
models = []
for seed in (0, 1):
    p = O.init_params(seed, True)
    m = sinnerf_amd.NeRF(use_new_activation=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    models.append(m.to(dev).train())
emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
rays = torch.from_numpy(O.lego_rays(400, 400, 0)[:: 160000 // N][:N]).to(dev)
tgt = torch.rand((N, 3), device=dev)


def step():
    for m in models:
        m.zero_grad(set_to_none=True)
    res = sinnerf_amd.render_rays(models, emb, rays, 64, False, 1.0, 1.0, 64, 32768, True)
    loss = ((res["rgb_fine"] - tgt) ** 2).mean() + ((res["rgb_coarse"] - tgt) ** 2).mean() + 0.1 * res["depth_fine"].mean()
    loss.backward()
    return loss


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flop = 3489024 * N * 192
out = {"rays": N, "ms_per_step": dt * 1e3, "train_rays_per_s": N / dt, "tflops_algorithmic": flop / dt / 1e12,
       "frac_fp32_mfma_peak": flop / dt / 157.3e12}

# stage breakdown on the fine pass (2/3 of the points)
m = models[1]
z = torch.sort(torch.rand((N, 128), device=dev) * 4 + 2, -1)[0].contiguous()
g = torch.randn((N, 128, 4), device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


P = N * 128
ms_fwd, raw = timed(lambda: A._MLPFn.apply(m, rays, z, *m.raw_tensors()))
acts, embt, outt = raw.grad_fn.saved_tensors if raw.grad_fn is not None else (None, None, None)
G = torch.empty((10, acts.shape[1], 256), device=dev); g_o = torch.empty((P, 4), device=dev)
from sinnerf_amd import _lib                           # noqa: E402
ms_chain, _ = timed(lambda: _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd()), 0, _lib.ptr(acts), _lib.ptr(outt),
                                                                      _lib.ptr(g), P, acts.shape[1], _lib.ptr(G), _lib.ptr(g_o), None), "chain"))
ms_dw, _ = timed(lambda: A._weight_grads(m, acts, embt, G, [True] * 24))
with torch.no_grad():
    ms_inf, _ = timed(lambda: sinnerf_amd.rendering._mlp(m, rays, z, False))
fl = lambda f: f * P / 1e9
out["fine_pass"] = {"points": P, "fwd_train_ms": ms_fwd, "fwd_infer_ms": ms_inf, "bwd_chain_ms": ms_chain, "dW_gemm_ms": ms_dw,
                    "fwd_train_tflops": fl(1186816) / ms_fwd, "bwd_chain_tflops": fl(2 * 569344) / ms_chain,
                    "dW_tflops": fl(1186816) / ms_dw}
# mixed precision: the same step with compute_dtype="bf16" (bf16-operand forward storing fp32 activations, fp32 backward)
mb = []
for seed in (0, 1):
    mm = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(seed, True).items()})
    mb.append(mm.to(dev).train())
models_fp32, models = models, mb
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dtb = (time.perf_counter() - t0) / K
ms_fwd_b, _ = timed(lambda: A._MLPFn.apply(mb[1], rays, z, *mb[1].raw_tensors()))
raw_b = A._MLPFn.apply(mb[1], rays, z, *mb[1].raw_tensors())
acts_b, emb_b, out_b = raw_b.grad_fn.saved_tensors
G_b = torch.zeros((10, acts_b.shape[1], 256), dtype=acts_b.dtype, device=dev)
ms_chain_b, _ = timed(lambda: _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(mb[1].packed_bwd("bf16")), 2, _lib.ptr(acts_b), _lib.ptr(out_b),
                                                                        _lib.ptr(g), P, acts_b.shape[1], _lib.ptr(G_b), _lib.ptr(g_o), None), "chain"))
ms_dw_b, _ = timed(lambda: A._weight_grads(mb[1], acts_b, emb_b, G_b, [True] * 24))
out["bf16_training"] = {"ms_per_step": dtb * 1e3, "train_rays_per_s": N / dtb, "fine_fwd_train_ms": ms_fwd_b,
                        "fine_bwd_chain_ms": ms_chain_b, "fine_dW_ms": ms_dw_b}
# fp32-level accuracy on the bf16 MFMA: compute_dtype="bf16x3" (3-term split forward, chain and weight gradients; the state as (hi, lo) pairs)
mx = []
for seed in (0, 1):
    mm = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16x3")
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(seed, True).items()})
    mx.append(mm.to(dev).train())
models = mx
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dtx = (time.perf_counter() - t0) / K
out["bf16x3_training"] = {"ms_per_step": dtx * 1e3, "train_rays_per_s": N / dtx, "tflops_algorithmic": flop / dtx / 1e12,
                          "x_fp32_mfma_peak": flop / dtx / 157.3e12}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/train_bench.json", "w"), indent=1)
print(json.dumps(out))
