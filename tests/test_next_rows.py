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


# ------------------------------------------------------------------------------------- losses on the rendered rays
LOSS_MODES = (("nomask", dict(use_mask=False)), ("gt0", dict(use_mask=True)), ("mask", dict(mask="mask")))


def test_oracle_render_loss_golden():
    """oracle MSELoss / SL1Loss / psnr restatement against the reference's own classes (tests/golden/loss.npz, made by
    oracle/gen_golden.py --loss from losses.py:12-22, models/sinnerf.py:32-42, metrics.py:5-15)."""
    z = np.load(f"{GOLDEN}/loss.npz")
    res = {k: z[k] for k in ("rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine")}
    for tag, kw in LOSS_MODES:
        kw = {k: (z["mask"] if isinstance(v, str) else v) for k, v in kw.items()}
        st, gr = O.render_loss(res, z["rgb_gt"], z["depth_gt"], w_depth=float(z["w_depth"]), **kw)
        assert abs(st["mse_coarse"] + st["mse_fine"] - z[f"{tag}_l2"]) <= 1e-6 * z[f"{tag}_l2"]
        for k in ("fine", "coarse"):
            assert abs(st["sl1_" + k] - z[f"{tag}_sl1_{k}"]) <= 1e-6 * z[f"{tag}_sl1_{k}"], (tag, k)
        assert abs(st["total"] - z[f"{tag}_total"]) <= 1e-6 * z[f"{tag}_total"]
        for k in res:
            assert np.abs(gr[k] - z[f"{tag}_g_{k}"]).max() <= 1e-6 * np.abs(z[f"{tag}_g_{k}"]).max(), (tag, k)
    st, _ = O.render_loss(res, z["rgb_gt"])
    assert abs(st["psnr_fine"] - z["psnr_fine"]) <= 1e-4 and abs(st["psnr_coarse"] - z["psnr_coarse"]) <= 1e-4
    assert abs(st["mse_fine"] - z["mse_fine"]) <= 1e-6 * z["mse_fine"]


@pytest.mark.gpu
def test_render_loss_gpu_golden_and_autograd():
    """sn_render_loss through the reference-shaped objects (MSELoss, SL1Loss, psnr) and the fused render_loss: values and
    gradients against the golden fixture; tolerance 2e-6 relative (fp32 values, double accumulation)."""
    from sinnerf_amd.losses import MSELoss, SL1Loss, psnr, render_loss
    z = np.load(f"{GOLDEN}/loss.npz")
    dev = torch.device("cuda:0")
    gt_rgb, gt_d = torch.from_numpy(z["rgb_gt"]).to(dev), torch.from_numpy(z["depth_gt"]).to(dev)
    w_depth = float(z["w_depth"])
    for tag, kw in LOSS_MODES:
        res = {k: torch.from_numpy(z[k]).to(dev).requires_grad_(True) for k in ("rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine")}
        rkw = dict(useMask=kw.get("use_mask", False), mask=torch.from_numpy(z["mask"]).to(dev) if "mask" in kw else None)
        total, st = render_loss(res, gt_rgb, gt_d, w_depth=w_depth, **rkw)
        total.backward()
        assert abs(total.item() - z[f"{tag}_total"]) <= 2e-6 * z[f"{tag}_total"], tag
        assert abs(st["mse_coarse"].item() + st["mse_fine"].item() - z[f"{tag}_l2"]) <= 2e-6 * z[f"{tag}_l2"]
        for k in ("fine", "coarse"):
            assert abs(st["sl1_" + k].item() - z[f"{tag}_sl1_{k}"]) <= 2e-6 * z[f"{tag}_sl1_{k}"], (tag, k)
        for k, v in res.items():
            ref = z[f"{tag}_g_{k}"]
            assert np.abs(v.grad.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max(), (tag, k)
        # the reference's separate objects, composed the way models/sinnerf.py:310-319 does
        res2 = {k: torch.from_numpy(z[k]).to(dev).requires_grad_(True) for k in res}
        skw = dict(useMask=kw.get("use_mask", False)) if "mask" not in kw else dict(mask=rkw["mask"])
        tot2 = MSELoss()(res2, gt_rgb)["tot"] + w_depth * (SL1Loss()(res2["depth_fine"], gt_d, **skw)
                                                          + SL1Loss()(res2["depth_coarse"], gt_d, **skw))
        tot2.backward()
        assert abs(tot2.item() - z[f"{tag}_total"]) <= 2e-6 * z[f"{tag}_total"]
        for k, v in res2.items():
            ref = z[f"{tag}_g_{k}"]
            assert np.abs(v.grad.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max(), (tag, k)
    assert abs(psnr(torch.from_numpy(z["rgb_fine"]).to(dev), gt_rgb).item() - z["psnr_fine"]) <= 1e-4
    # only-coarse model (N_importance = 0): missing keys drop their terms (losses.py:19)
    only_c = {"rgb_coarse": torch.from_numpy(z["rgb_coarse"]).to(dev)}
    assert abs(MSELoss()(only_c, gt_rgb)["tot"].item() - O.mse_loss(z["rgb_coarse"], z["rgb_gt"])) <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 160000])
def test_render_loss_gpu_sizes_vs_oracle(n):
    from sinnerf_amd.losses import render_loss
    r = np.random.RandomState(n)
    gt_rgb, gt_d = r.uniform(0, 1, (n, 3)).astype(np.float32), r.uniform(0.5, 6, n).astype(np.float32)
    res = {"rgb_coarse": (gt_rgb + r.normal(0, 0.1, (n, 3))).astype(np.float32), "rgb_fine": (gt_rgb + r.normal(0, 0.03, (n, 3))).astype(np.float32),
           "depth_coarse": (gt_d + r.normal(0, 2, n)).astype(np.float32), "depth_fine": (gt_d + r.normal(0, 0.5, n)).astype(np.float32)}
    st, gr = O.render_loss(res, gt_rgb, gt_d, w_rgb=1.0, w_depth=0.1)
    dev = torch.device("cuda:0")
    tres = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in res.items()}
    total, got = render_loss(tres, torch.from_numpy(gt_rgb).to(dev), torch.from_numpy(gt_d).to(dev), w_depth=0.1)
    (3.0 * total).backward()
    assert abs(total.item() - st["total"]) <= 2e-6 * st["total"]
    for k in ("mse_coarse", "mse_fine", "sl1_coarse", "sl1_fine", "psnr_fine"):
        assert abs(got[k].item() - st[k]) <= 2e-6 * abs(st[k]) + 1e-5 * (k == "psnr_fine"), k
    for k, v in tres.items():
        assert np.abs(v.grad.cpu().numpy() - 3.0 * gr[k]).max() <= 3e-6 * np.abs(3.0 * gr[k]).max(), k


# ------------------------------------------------------------------------------------- eval-side formats / callers
def test_pfm_bytes_match_reference(tmp_path):
    """save_pfm writes byte-for-byte what datasets/depth_utils.py:46-74 writes (tests/golden/pfm.npz, gen_golden.py --pfm)
    and read_pfm returns what :6-43 returns."""
    from sinnerf_amd.evalio import read_pfm, save_pfm
    z = np.load(f"{GOLDEN}/pfm.npz")
    for name, scale in (("grey", 1), ("color", 2.5)):
        path = os.path.join(tmp_path, name + ".pfm")
        save_pfm(path, z[name], scale)
        assert np.array_equal(np.frombuffer(open(path, "rb").read(), np.uint8), z[name + "_bytes"]), name
        back, sc = read_pfm(path)
        assert np.array_equal(back, z[name + "_read"]) and np.array_equal(back, z[name]) and sc == float(z[name + "_scale"])
    with pytest.raises(Exception):
        save_pfm(os.path.join(tmp_path, "x.pfm"), z["grey"].astype(np.float64))
    open(os.path.join(tmp_path, "bad.pfm"), "wb").write(b"P6\n1 1\n255\n")
    with pytest.raises(Exception):
        read_pfm(os.path.join(tmp_path, "bad.pfm"))


def test_png_writer_roundtrip(tmp_path):
    """to_uint8 = eval.py:182; the PNG is decoded back with a stdlib-only reader (zlib + filter 0)."""
    import struct
    import zlib
    from sinnerf_amd.evalio import save_png, to_uint8
    img = np.random.RandomState(0).uniform(-0.001, 1.001, (9, 13, 3)).astype(np.float32)
    u8 = to_uint8(np.clip(img, 0, 1))
    assert np.array_equal(u8, (np.clip(img, 0, 1) * 255).astype(np.uint8))
    path = os.path.join(tmp_path, "a.png")
    save_png(path, u8)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert hdr == (13, 9, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(9, 1 + 13 * 3)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(9, 13, 3), u8)


@pytest.mark.gpu
def test_render_frame_eval_driver_vs_oracle(tmp_path):
    """eval.py:152-189 for one pose through render_frame: GPU ray generation + batched_inference; image / depth against the
    oracle render of the same rays; files written in the reference's formats."""
    import sinnerf_amd
    from sinnerf_amd.evalio import read_pfm, render_frame, save_pfm, save_png, to_uint8
    from tests.helpers import check_render
    from tests.test_parity_gpu import embeddings, make_model
    mc, pc = make_model(0, True)
    mf, pf = make_model(1, True)
    H, W = 20, 30
    rays_np = O.lego_rays(H, W, seed=3)                                      # radius-4 pose looking at the origin
    ref = O.render_rays([pc, pf], rays_np, 64, False, 0, 0, 64, 1 << 19, True, False)
    res = sinnerf_amd.evalio.batched_inference([mc, mf], embeddings(), torch.from_numpy(rays_np).cuda(), 64, 64, False, 1024, True)
    check_render({k: v.cpu().numpy() for k, v in res.items()}, ref, tag="batched_inference")
    img = res["rgb_fine"].view(H, W, 3).cpu().numpy()
    depth = np.nan_to_num(res["depth_fine"].view(H, W).cpu().numpy())
    save_png(os.path.join(tmp_path, "000.png"), to_uint8(np.clip(img, 0, 1)))
    save_pfm(os.path.join(tmp_path, "depth_000.pfm"), depth)
    back, _ = read_pfm(os.path.join(tmp_path, "depth_000.pfm"))
    assert np.array_equal(back, depth)
    # render_frame = the same with rays generated on the GPU from (c2w, focal)
    z = np.load(f"{GOLDEN}/rays.npz")
    Hh, Ww = int(z["H"]), int(z["W"])
    img2, depth2, res2 = render_frame([mc, mf], embeddings(), z["c2w"], Hh, Ww, float(z["focal"]), float(z["near"]), float(z["far"]))
    ref2 = O.render_rays([pc, pf], z["rays"], 64, False, 0, 0, 64, 1 << 19, True, False)
    assert img2.shape == (Hh, Ww, 3) and depth2.shape == (Hh, Ww)
    assert np.abs(img2.reshape(-1, 3) - ref2["rgb_fine"]).max() <= 2e-3
