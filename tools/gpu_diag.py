#!/usr/bin/env python3
"""GPU diagnostics for the backward path: per-slot comparison of stored activations / chain gradients with the
oracle on a ragged point count.  Writes gpurun_out/diag.json."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import _lib                           # noqa: E402

dev = torch.device("cuda:0")
out = {}
p = O.init_params(3, True)
m = sinnerf_amd.NeRF(use_new_activation=True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
m = m.to(dev)
for (n, S) in ((60, 37), (64, 64)):
    rays = O.lego_rays(400, 400, seed=0)[::2503][:n]
    z = O.coarse_z_vals(rays, S, False, 1.0, np.random.RandomState(1).uniform(0, 1, (n, S)).astype(np.float32))
    g = np.random.RandomState(2).standard_normal((n, S, 4)).astype(np.float32)
    P = n * S
    rays_t, z_t, g_t = (torch.from_numpy(a).to(dev) for a in (rays, z, g))
    raw = torch.empty((n, S, 4), device=dev); acts = torch.full((10, P, 256), float("nan"), device=dev)
    emb = torch.zeros((P, 128), device=dev)
    _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(m.packed()), 0, _lib.ptr(rays_t), _lib.ptr(z_t), n, S, _lib.ptr(raw),
                                             _lib.ptr(acts), _lib.ptr(emb), P, None), "fwd")
    G = torch.full((10, P, 256), float("nan"), device=dev); g_o = torch.full((P, 4), float("nan"), device=dev)
    _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd()), 0, _lib.ptr(acts), _lib.ptr(raw), _lib.ptr(g_t), P, P,
                                              _lib.ptr(G), _lib.ptr(g_o), None), "bwd")
    torch.cuda.synchronize()
    xin = np.concatenate([O.embedding(O._points(rays, z).reshape(-1, 3), 10), np.repeat(O.embedding(rays[:, 3:6], 4), S, 0)], 1)
    cache, gy = {}, {}
    O.nerf_forward(p, xin, cache=cache)
    O.nerf_backward(p, cache, g.reshape(-1, 4), gy_out=gy)
    ref_act = {i: cache[f"h{i+1}"] for i in range(8)}; ref_act[8] = cache["final"]; ref_act[9] = cache["d"]
    ref_g = {i: gy[f"l{i+1}"] for i in range(8)}; ref_g[8] = gy["final"]; ref_g[9] = gy["dir"]
    res = {}
    A, Gn = acts.cpu().numpy(), G.cpu().numpy()
    for slot in range(10):
        w = ref_act[slot].shape[1]
        da = np.abs(A[slot][:, :w] - ref_act[slot]); dg = np.abs(Gn[slot][:, :w] - ref_g[slot])
        res[f"slot{slot}"] = {"act_max_err": float(np.nanmax(da)), "act_nan": int(np.isnan(A[slot][:, :w]).sum()),
                              "act_worst_row": int(np.nanargmax(da.max(1))), "g_max_err": float(np.nanmax(dg)),
                              "g_nan": int(np.isnan(Gn[slot][:, :w]).sum()), "g_worst_row": int(np.nanargmax(dg.max(1))),
                              "g_scale": float(np.abs(ref_g[slot]).max()),
                              "g_bad_rows": np.nonzero(dg.max(1) > 1e-4 * np.abs(ref_g[slot]).max())[0][:20].tolist()}
    e = emb.cpu().numpy()
    res["emb_xyz_err"] = float(np.nanmax(np.abs(e[:, :63] - xin[:, :63]))); res["emb_dir_err"] = float(np.nanmax(np.abs(e[:, 64:91] - xin[:, 63:])))
    res["emb_nan"] = int(np.isnan(e).sum())
    res["g_o_err"] = float(np.abs(g_o.cpu().numpy() - np.concatenate([gy["rgb"], gy["sigma"]], 1)).max())
    out[f"n{n}_S{S}"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
