"""SURVEY §8f 'next' rows: on-GPU ray generation, flat Adam, checkpoint compatibility."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as O
from tests.helpers import GOLDEN


def test_oracle_get_rays_golden():
    z = np.load(f"{GOLDEN}/rays.npz")
    got = O.get_rays(int(z["H"]), int(z["W"]), float(z["focal"]), z["c2w"], float(z["near"]), float(z["far"]))
    assert got.shape == z["rays"].shape
    assert np.abs(got - z["rays"]).max() <= 2e-6


def test_checkpoint_helpers_roundtrip(tmp_path):
    """A Lightning-style SinNeRF checkpoint (prefixed keys + foreign modules) loads into sinnerf_amd.NeRF through the
    reference's helper signatures (utils/__init__.py:60-83, eval.py:139-140)."""
    from sinnerf_amd import NeRF
    from sinnerf_amd.ckpt import extract_model_state_dict, load_ckpt, save_weights_only
    pc, pf = O.init_params(0, True), O.init_params(1, True)
    sd = {"nerf_coarse." + k: torch.from_numpy(v) for k, v in pc.items()}
    sd.update({"nerf_fine." + k: torch.from_numpy(v) for k, v in pf.items()})
    sd["discriminator.conv.weight"] = torch.zeros(3)               # foreign keys must be skipped
    path = os.path.join(tmp_path, "epoch=3.ckpt")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    m = NeRF(use_new_activation=True)
    load_ckpt(m, path, model_name="nerf_fine")
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), pf[k]), k
    assert set(extract_model_state_dict(path, "nerf_coarse").keys()) == set(pc.keys())
    assert "xyz_encoding_1.0.bias" not in extract_model_state_dict(path, "nerf_coarse", prefixes_to_ignore=["xyz_encoding_1"])
    path2 = os.path.join(tmp_path, "weights.ckpt")
    save_weights_only(sd, path2)                                    # utils/save_weights_only.py
    m2 = NeRF(use_new_activation=True)
    load_ckpt(m2, path2, model_name="nerf_coarse")
    assert np.array_equal(m2.state_dict()["rgb.0.weight"].numpy(), pc["rgb.0.weight"])


@pytest.mark.gpu
def test_generate_rays_gpu_golden_and_window():
    from sinnerf_amd.ray_utils import get_rays
    z = np.load(f"{GOLDEN}/rays.npz")
    H, W = int(z["H"]), int(z["W"])
    c2w = torch.from_numpy(z["c2w"]).cuda()
    rays = get_rays(H, W, float(z["focal"]), c2w, float(z["near"]), float(z["far"])).cpu().numpy()
    assert rays.shape == z["rays"].shape and np.abs(rays - z["rays"]).max() <= 2e-6
    win = (3, 2, 4, 3, 7, 5)                                        # x0, y0, sx, sy, pw, ph
    sub = get_rays(H, W, float(z["focal"]), c2w, float(z["near"]), float(z["far"]), window=win).cpu().numpy()
    ref = z["rays"].reshape(H, W, 8)[2:2 + 5 * 3:3, 3:3 + 7 * 4:4].reshape(-1, 8)
    assert np.abs(sub - ref).max() <= 2e-6
    from sinnerf_amd._lib import SinnerfHipError
    with pytest.raises(SinnerfHipError):
        get_rays(H, W, 40.0, c2w, 2.0, 6.0, window=(30, 0, 4, 1, 7, 5))     # window leaves the image


@pytest.mark.gpu
def test_flat_adam_matches_torch_adam():
    from sinnerf_amd import NeRF
    from sinnerf_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = [NeRF(use_new_activation=True).to(dev), NeRF(use_new_activation=True).to(dev)]
    b = [NeRF(use_new_activation=True).to(dev), NeRF(use_new_activation=True).to(dev)]
    for x, y in zip(a, b):
        y.load_state_dict(x.state_dict())
    opt_a = FlatAdam(a, lr=5e-4, eps=1e-8, weight_decay=1e-3)
    opt_b = torch.optim.Adam([p for m in b for p in m.parameters()], lr=5e-4, eps=1e-8, weight_decay=1e-3)
    assert opt_a.flat.numel() == 1191688
    g = torch.Generator(device=dev).manual_seed(1)
    for step in range(5):
        opt_a.zero_grad(); opt_b.zero_grad(set_to_none=False)
        for pa, pb in zip(opt_a.grads.params, [p for m in b for p in m.parameters()]):
            d = torch.randn(pa.shape, device=dev, generator=g) * 0.01
            pa.grad.add_(d)
            pb.grad = d.clone() if pb.grad is None else pb.grad.copy_(d)
        opt_a.step(); opt_b.step()
    for x, y in zip(a, b):
        for (k, va), (_, vb) in zip(x.state_dict().items(), y.state_dict().items()):
            assert torch.allclose(va, vb, rtol=1e-5, atol=1e-7), k
    # the models still render through the packed-weight cache after the in-place update
    import sinnerf_amd
    rays = torch.from_numpy(O.lego_rays(400, 400, 0)[::4000]).to(dev)
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    with torch.no_grad():
        ra = sinnerf_amd.render_rays(a, emb, rays, 64, False, 0, 0, 64, 32768, True)["rgb_fine"]
        rb = sinnerf_amd.render_rays(b, emb, rays, 64, False, 0, 0, 64, 32768, True)["rgb_fine"]
    assert torch.allclose(ra, rb, atol=1e-5)
