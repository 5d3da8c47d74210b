"""Where does the bf16 MLP kernel disagree with the fp32 one?  (debug aid)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_parity_gpu import make_model, dev, O
from sinnerf_amd import rendering
mb, _ = make_model(0, True, dtype="bf16")
mf, _ = make_model(0, True)
for n_rays, S in ((100, 70), (1000, 64), (333, 128)):
    rays = O.lego_rays(400, 400, seed=0)[:: 160000 // n_rays][:n_rays]
    z = O.coarse_z_vals(rays, S, False, 1.0, np.random.RandomState(1).uniform(0, 1, (rays.shape[0], S)).astype(np.float32))
    r = torch.from_numpy(rays).to(dev()); zz = torch.from_numpy(z).to(dev())
    for so in (False, True):
        with torch.no_grad():
            a = rendering._mlp(mf, r, zz, so).cpu().numpy().reshape(rays.shape[0] * S, -1)
            outs = [rendering._mlp(mb, r, zz, so).cpu().numpy().reshape(rays.shape[0] * S, -1) for _ in range(3)]
        for k, b in enumerate(outs):
            err = np.abs(a - b)
            badrows = np.unique(np.where(~(err < 0.05))[0])
            print(f"P={rays.shape[0]*S} sigma_only={so} run{k}: max err {np.nanmax(err):.3g} bad rows {len(badrows)} {badrows[:12]} same_as_run0={np.array_equal(b, outs[0], equal_nan=True)}")
            if len(badrows):
                i = badrows[0]; print("   row", i, "tile", i // 256, "wave", (i % 256) // 64, "pt", (i % 64) // 32, "j", i % 32, b[i], a[i])
