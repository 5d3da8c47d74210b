#!/usr/bin/env python3
"""Run-to-run comparison of a training render (llff patch, 5 292 rays, perturb = noise_std = 1, injected draws) per arithmetic, with freed
memory poisoned in between: bit-identical outputs and gradients or not, and where they differ.  This is what found the store-order race of
the bf16x3 training kernels (csrc/sn_mlp_x3.h x3_store_step); tests/test_bf16x3_gpu.py::test_training_render_is_run_to_run_identical is the
regression test.  usage (repo root = $R): R=$PWD python tools/x3_determinism.py     (SINNERF_HIP_LIB=... for another build)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("R", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_np as O
import sinnerf_amd
from tests.test_parity_gpu import make_model, embeddings, injected_rng, dev
rays = O.llff_patch_rays(0)
n, S, NI = rays.shape[0], 64, 64
r = np.random.RandomState(11)
rng = {"perturb": r.uniform(0, 1, (n, S)).astype(np.float32), "noise_coarse": r.standard_normal((n, S)).astype(np.float32),
       "u": r.uniform(0, 1, (n, NI)).astype(np.float32), "noise_fine": r.standard_normal((n, S + NI)).astype(np.float32)}
coef = {k: torch.from_numpy(r.standard_normal(sh).astype(np.float32) / n).to(dev()) for k, sh in (("rgb_coarse", (n, 3)), ("rgb_fine", (n, 3)), ("depth_coarse", (n,)), ("depth_fine", (n,)))}
order = [("rand", rng["perturb"]), ("randn", rng["noise_coarse"]), ("rand", rng["u"]), ("randn", rng["noise_fine"])]
def run(dt, junk=None):
    mc, _ = make_model(0, True, dtype=dt); mf, _ = make_model(1, True, dtype=dt)
    mc.train(); mf.train()
    with injected_rng(order) as left:
        res = sinnerf_amd.render_rays([mc, mf], embeddings(), torch.from_numpy(rays).to(dev()), S, False, 1.0, 1.0, NI, 32768, False)
    sum((res[k] * v).sum() for k, v in coef.items()).backward()
    torch.cuda.synchronize()
    return [{k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()} for m in (mc, mf)], {k: v.detach().cpu().numpy() for k, v in res.items()}
for dt in ("bf16x3", "fp32"):
    a, ra = run(dt)
    junk = torch.full((200, 1024, 1024), float("nan"), device=dev()); del junk          # poison freed memory
    b, rb = run(dt)
    same_f = all(np.array_equal(ra[k], rb[k]) for k in ra)
    worst = max(np.abs(a[i][k] - b[i][k]).max() / (np.abs(a[i][k]).max() + 1e-30) for i in range(2) for k in a[i])
    bad = [(i, k) for i in range(2) for k in a[i] if not np.array_equal(a[i][k], b[i][k])]
    print(dt, "forward identical:", same_f, "| gradients identical:", not bad, "| worst rel diff %.2e" % worst, bad[:4])
# where: per output, how many rays differ between two runs of the same arithmetic (bf16x3), under autograd and under no_grad
def fw(dt, grad):
    mc, _ = make_model(0, True, dtype=dt); mf, _ = make_model(1, True, dtype=dt)
    (mc.train(), mf.train()) if grad else (mc.eval(), mf.eval())
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx, injected_rng(order) as left:
        res = sinnerf_amd.render_rays([mc, mf], embeddings(), torch.from_numpy(rays).to(dev()), S, False, 1.0, 1.0, NI, 32768, False)
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in res.items()}
for grad in (False, True):
    x = fw("bf16x3", grad); junk = torch.full((200, 1024, 1024), float("nan"), device=dev()); del junk; y = fw("bf16x3", grad)
    for k in x:
        d = np.abs(x[k] - y[k]).reshape(n, -1).max(1)
        print("autograd" if grad else "no_grad ", k, "rays differing", int((d > 0).sum()), "max diff %.3e" % d.max(), "first rows", np.nonzero(d > 0)[0][:6])
