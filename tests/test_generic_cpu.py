"""sinnerf_amd/generic.py (the reference's GENERAL configurations as torch ops; on the product path it only ever sees device tensors:
NeRF.forward / render_rays refuse CPU tensors before they get here) is device-agnostic torch code, so its arithmetic is pinned on the
CPU against the numpy oracle: another layer configuration, disparity sampling, stratified perturbation and noise with the draws
injected in the reference's consumption order."""
import contextlib

import numpy as np
import torch

from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import generic


@contextlib.contextmanager
def injected(order):
    q = list(order)
    real_rand, real_randn = torch.rand, torch.randn

    def take(kind, shape):
        k, arr = q.pop(0)
        assert k == kind and tuple(arr.shape) == tuple(shape), (k, kind, arr.shape, shape)
        return torch.from_numpy(arr)
    torch.rand = lambda *a, **kw: take("rand", a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a)
    torch.randn = lambda *a, **kw: take("randn", a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a)
    try:
        yield q
    finally:
        torch.rand, torch.randn = real_rand, real_randn


def test_general_layer_configuration_matches_oracle():
    torch.manual_seed(0)
    for kw, okw in ((dict(D=4, W=128, skips=[2]), dict(D=4, W=128, skips=(2,))), (dict(D=6, W=64, skips=[2, 4]), dict(D=6, W=64, skips=(2, 4)))):
        m = sinnerf_amd.NeRF(use_new_activation=True, **kw)
        assert not m.fused
        params = {k: v.detach().numpy() for k, v in m.state_dict().items()}
        x = np.random.RandomState(0).standard_normal((200, 90)).astype(np.float32)
        got = generic.mlp_generic(m, torch.from_numpy(x)).detach().numpy()
        assert np.abs(got - O.nerf_forward(params, x, **okw)).max() <= 1e-5
        sig = generic.mlp_generic(m, torch.from_numpy(x[:, :63]), sigma_only=True).detach().numpy()
        assert np.abs(sig - O.nerf_forward(params, x[:, :63], sigma_only=True, **okw)).max() <= 1e-5


def test_general_render_matches_oracle_with_injected_draws():
    pc, pf = O.init_params(0, True), O.init_params(1, True)
    mc, mf = sinnerf_amd.NeRF(use_new_activation=True), sinnerf_amd.NeRF(use_new_activation=True)
    mc.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
    mf.load_state_dict({k: torch.from_numpy(v) for k, v in pf.items()})
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    rays = O.lego_rays(400, 400, seed=0)[::2503][:40]
    n = rays.shape[0]
    r = np.random.RandomState(5)
    for use_disp, white_back, test_time in ((True, True, False), (False, False, True)):
        rng = {"perturb": r.uniform(0, 1, (n, 24)).astype(np.float32), "noise_coarse": r.standard_normal((n, 24)).astype(np.float32),
               "u": r.uniform(0, 1, (n, 40)).astype(np.float32), "noise_fine": r.standard_normal((n, 64)).astype(np.float32)}
        order = [("rand", rng["perturb"]), ("randn", rng["noise_coarse"]), ("rand", rng["u"]), ("randn", rng["noise_fine"])]
        with injected(order) as left, torch.no_grad():
            res = generic.render_generic([mc, mf], emb, torch.from_numpy(rays), 24, use_disp, 1.0, 0.5, 40, 4096, white_back, test_time, False)
            assert not left
        ref = O.render_rays([pc, pf], rays, 24, use_disp, 1.0, 0.5, 40, 1 << 19, white_back, test_time, rng=rng)
        assert set(res) == set(ref)
        for k in ref:
            a, b = res[k].numpy(), ref[k]
            tol = 2e-4 if k.startswith("opacity") else 1e-3 * np.abs(b).max() + 1e-5
            # (a sample within rounding of a cdf knot may land one bin away: bound the fraction, not every entry)
            assert (np.abs(a - b) > tol).mean() <= 0.01, (k, np.abs(a - b).max())
