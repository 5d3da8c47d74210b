#!/usr/bin/env python3
"""One-shot GPU diagnostics (run under gpurun): stage-by-stage error statistics of the HIP path against the
oracle, written to gpurun_out/diag.json -- more informative than a failing assert when no GPU is at hand."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import _lib, rendering               # noqa: E402

dev = torch.device("cuda:0")
out = {"device": torch.cuda.get_device_name(0)}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def stats(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    return {"max_abs": float(d.max()), "max_rel": float((d / (np.abs(ref) + 1e-3)).max()), "mean_abs": float(d.mean()),
            "nan": int(np.isnan(got).sum()), "ref_absmax": float(np.abs(ref).max())}


p = O.init_params(0, True)
m = sinnerf_amd.NeRF(use_new_activation=True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
m = m.to(dev).eval()
rays = O.lego_rays(400, 400, 0)[::1601][:100]
z = O.coarse_z_vals(rays, 64, False, 0, None)
for flags in (0, 1):
    for so in (False, True):
        ref = O._run_model(p, rays, z, O.embedding(rays[:, 3:6], 4), so, 1 << 20)
        try:
            with torch.no_grad():
                got = rendering._mlp(m, t(rays), t(z), so, flags).cpu().numpy()
            torch.cuda.synchronize()
            out[f"mlp_flags{flags}_sigma{int(so)}"] = stats(got, ref)
            if not so:
                out[f"mlp_flags{flags}_percol"] = [stats(got[..., c], ref[..., c]) for c in range(4)]
        except Exception as e:                          # noqa: BLE001
            out[f"mlp_flags{flags}_sigma{int(so)}"] = {"error": repr(e)}

# embedded-input path: per-layer insight is not available, but the first rows help when the layout is wrong
xin = np.concatenate([O.embedding(O._points(rays, z).reshape(-1, 3), 10), np.repeat(O.embedding(rays[:, 3:6], 4), 64, 0)], 1)
with torch.no_grad():
    got = m(t(xin)).cpu().numpy()
ref = O.nerf_forward(p, xin)
out["mlp_embedded"] = stats(got, ref)
out["mlp_embedded_first_rows"] = {"got": got[:3].tolist(), "ref": ref[:3].tolist()}

# timing of the MLP kernel alone at frame size
rays_f = t(O.lego_rays(400, 400, 0))
zf = torch.empty((160000, 128), device=dev)
_lib.check(_lib.lib.sn_sample_coarse(_lib.ptr(rays_f), 160000, 128, 0, 0.0, None, _lib.ptr(zf), None), "coarse")
for flags in (0, 1):
    with torch.no_grad():
        rendering._mlp(m, rays_f, zf, False, flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rendering._mlp(m, rays_f, zf, False, flags)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out[f"mlp_frame_fine_flags{flags}"] = {"ms": dt * 1e3, "tflops": 1186816 * 160000 * 128 / dt / 1e12}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
