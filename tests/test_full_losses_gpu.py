"""BASELINE configs[2] as it is named -- "llff/room 504x378 patch 63x84, full SinNeRF losses" -- on the part that is runnable offline:
the adversarial term (models/sinnerf.py:143-145, 445-471, hinge) through the reference's UNMODIFIED ``models/discriminator.py``
(staged byte-for-byte into oracle/_ref by build(); oracle/stage_ref.py), plugged into ``SinNeRFSystem`` where the reference builds it.
The DINO-ViT term (sinnerf.py:332-339) needs weights from the network: not runnable offline, not built.

What is checked: the gradient of a conv discriminator's output flowing back THROUGH the HIP render into the NeRF parameters equals the
gradient torch autograd derives through the staged reference ``render_rays`` + reference ``NeRF`` under the same discriminator, the same
random draws and the same augmentation draws; and the second optimiser pass (optimizer_idx = 1, sinnerf.py:462-471)."""
import copy
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_np as O                                                     # noqa: E402
from oracle import stage_ref                                                          # noqa: E402
from tests.test_parity_gpu import dev, embeddings, injected_rng, make_model          # noqa: E402

PSX, PSY = 63, 84                                                                     # --patch_size_x 63 --patch_size_y 84 (README, LLFF)


def _discriminator(seed=0):
    if not stage_ref.discriminator_available():
        if stage_ref.stage() is None or not stage_ref.discriminator_available():
            pytest.skip("oracle/_ref does not hold models/discriminator.py (build() stages it where /root/reference exists)")
    mod = stage_ref.load_discriminator()
    torch.manual_seed(seed)
    # sinnerf.py:144-145 with --patch_size unset (-1) as in the LLFF commands: the generic branch, logits (1, 1, 12, 18) for a 63 x 84 patch
    return mod.Discriminator(conditional=False, policy="color,cutout", imsize=-1).to(dev())


def _draws(n, S, NI, seed):
    r = np.random.RandomState(seed)
    return [("rand", r.uniform(0, 1, (n, S)).astype(np.float32)), ("randn", r.standard_normal((n, S)).astype(np.float32)),
            ("rand", r.uniform(0, 1, (n, NI)).astype(np.float32)), ("randn", r.standard_normal((n, S + NI)).astype(np.float32))]


import contextlib


@contextlib.contextmanager
def injected_then_real(order):
    """the recorded draws for the render (in consumption order), the real generators afterwards (the discriminator's DiffAugment
    draws with torch.rand too, behind the render, inside the same call)"""
    q = list(order)
    real_rand, real_randn = torch.rand, torch.randn

    def make(kind, real):
        def f(*a, **kw):
            if not q:
                return real(*a, **kw)
            k, arr = q.pop(0)
            shape = a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a
            assert k == kind and tuple(arr.shape) == tuple(shape), (k, kind, arr.shape, shape)
            return torch.from_numpy(arr).to(dev())
        return f
    torch.rand, torch.randn = make("rand", real_rand), make("randn", real_randn)
    try:
        yield q
    finally:
        torch.rand, torch.randn = real_rand, real_randn


def _d_of_patch(D, rgb_fine):
    """D on the rendered side patch with REPRODUCIBLE augmentation draws (Discriminator.forward: np.random.random() > 0.5 ->
    DiffAugment 'color,cutout', torch.rand / torch.randint inside)"""
    np.random.seed(5)
    torch.manual_seed(5)
    return D(rgb_fine.reshape(1, PSX, PSY, 3).permute(0, 3, 1, 2))


def test_discriminator_gradient_through_the_hip_render_equals_the_reference_path():
    import sinnerf_amd
    rays_np = O.llff_patch_rays(0)
    n, S, NI = rays_np.shape[0], 64, 64
    assert n == PSX * PSY
    rays = torch.from_numpy(rays_np).to(dev())
    D0 = _discriminator()
    # ---- the HIP path (fp32) ----
    mc, pc = make_model(0, True)
    mf, pf = make_model(1, True)
    mc.train(); mf.train()
    Da = copy.deepcopy(D0)                       # spectral norm's power iteration updates buffers on every training-mode call
    with injected_rng(_draws(n, S, NI, 21)) as left:
        res = sinnerf_amd.render_rays([mc, mf], embeddings(), rays, S, False, 1.0, 1.0, NI, 32768, False)
        assert not left
    loss_a = -torch.mean(_d_of_patch(Da, res["rgb_fine"]))
    loss_a.backward()
    got = [{k: (torch.zeros_like(p) if p.grad is None else p.grad).detach().double().cpu().numpy() for k, p in m.named_parameters()} for m in (mc, mf)]
    # ---- the staged reference path on the same device: reference render_rays + reference NeRF, torch autograd ----
    ref_rendering, _ = stage_ref.load()
    ref_models, ref_emb = stage_ref.build_reference_models([pc, pf])
    ref_models = [m.to(dev()).train() for m in ref_models]
    Db = copy.deepcopy(D0)
    with injected_rng(_draws(n, S, NI, 21)) as left:
        ref = ref_rendering.render_rays(ref_models, ref_emb, rays, S, False, 1.0, 1.0, NI, 32768, False)
        assert not left
    loss_b = -torch.mean(_d_of_patch(Db, ref["rgb_fine"]))
    loss_b.backward()
    want = [{k: (torch.zeros_like(p) if p.grad is None else p.grad).detach().double().cpu().numpy() for k, p in m.named_parameters()} for m in ref_models]
    assert abs(loss_a.item() - loss_b.item()) <= 1e-4 * max(1.0, abs(loss_b.item())), (loss_a.item(), loss_b.item())
    d_rgb = (res["rgb_fine"].detach() - ref["rgb_fine"].detach()).abs().max().item()
    assert d_rgb <= 1e-4, d_rgb
    # only rgb_fine carries the loss: the coarse network receives no gradient on either path (z_fine is detached, rendering.py:312)
    assert all(p.grad is None or not p.grad.any() for p in ref_models[0].parameters())
    assert all(not np.any(v) for v in got[0].values())
    worst, wcos = 0.0, 1.0
    for k, v in want[1].items():
        g = got[1][k]
        nv = np.linalg.norm(v)
        assert nv > 0, k
        e = np.linalg.norm(g - v) / nv
        c = float((g * v).sum() / (np.linalg.norm(g) * nv))
        worst, wcos = max(worst, e), min(wcos, c)
        # the fine-network bar of the golden gradient tests (tests/test_grads_gpu.py: 5e-3; ReLU kinks between two fp32 evaluations)
        assert e <= 5e-3 and c >= 0.9999, (k, e, c)
    # the discriminator's own parameter gradients agree too (same patch to 1e-4, same augmentation)
    for (ka, pa), (kb, pb) in zip(Da.named_parameters(), Db.named_parameters()):
        assert ka == kb
        e = (pa.grad - pb.grad).norm().item() / max(pb.grad.norm().item(), 1e-30)
        assert e <= 5e-3, (ka, e)
    print("D -> HIP render -> NeRF gradients vs the staged reference path: worst norm-wise %.2e, min cosine %.7f; |d rgb_fine| %.1e; loss %.6f / %.6f"
          % (worst, wcos, d_rgb, loss_a.item(), loss_b.item()))


def test_both_optimizer_passes_of_the_patch_step():
    """SinNeRFSystem with the reference discriminator attached: configure_optimizers returns (opt, opt_d) (sinnerf.py:202-210);
    pass 0 moves the NeRFs and leaves D untouched and gradient-free; pass 1 (optimizer_idx = 1) is the hinge discriminator loss of
    sinnerf.py:465-471 on the real patch and the DETACHED side render -- equal to the formula evaluated on the reference render."""
    from sinnerf_amd.system import SinNeRFSystem
    D = _discriminator(1)
    torch.manual_seed(3)
    sysm = SinNeRFSystem(N_importance=64, perturb=1.0, noise_std=1.0, white_back=False, depth_weight=1.0, lr=5e-4, dis_weight=0.01).to(dev())
    sysm.attach_discriminator(D, patch_hw=(PSX, PSY))
    opts, scheds = sysm.configure_optimizers()
    assert len(opts) == 2 and opts[1] is sysm.opt_d and abs(opts[1].param_groups[0]["lr"] - 0.2 * 5e-4) < 1e-12
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    g = torch.Generator().manual_seed(0)
    u = lambda *sh: torch.rand(*sh, generator=g).to(dev())
    full, side = O.llff_patch_rays(1), O.llff_patch_rays(2)
    rnd, proj = O.llff_like_rays(4096, 0), O.llff_like_rays(4096, 3)
    batch = {"rays": t(rnd), "rgbs": u(4096, 3), "depth": 1.2 + 6.8 * u(4096), "rays_full": t(full), "rgbs_full": u(full.shape[0], 3),
             "rays_side": t(side), "rays_proj": t(proj), "depth_proj": 1.2 + 6.8 * u(4096), "real_patch": u(1, 3, PSX, PSY)}
    w0 = [p.detach().clone() for p in sysm.nerf_fine.parameters()]
    d0 = [p.detach().clone() for p in D.parameters()]
    # pass 0 alone: generator step with D frozen
    for p in D.parameters():
        p.requires_grad_(False)
    out = sysm.train_step(batch)
    for p in D.parameters():
        p.requires_grad_(True)
    assert torch.isfinite(out["loss"]).item()
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, sysm.nerf_fine.parameters()))
    assert all(torch.equal(a, b.detach()) for a, b in zip(d0, D.parameters())) and all(p.grad is None for p in D.parameters())
    # pass 1: the loss the reference forms, on a reference render of the same side rays with the same draws
    n = side.shape[0]
    Dc = copy.deepcopy(D)
    np.random.seed(9); torch.manual_seed(9)
    with injected_then_real(_draws(n, 64, 64, 4)) as left:
        out_d = sysm.training_step(batch, 0, optimizer_idx=1)
        assert not left
    ref_rendering, _ = stage_ref.load()
    params = [{k: v.detach().cpu().numpy() for k, v in m.state_dict().items()} for m in sysm.models]
    ref_models, ref_emb = stage_ref.build_reference_models(params)
    ref_models = [m.to(dev()) for m in ref_models]
    with torch.no_grad(), injected_rng(_draws(n, 64, 64, 4)) as left:
        rs = ref_rendering.render_rays(ref_models, ref_emb, t(side), 64, False, 1.0, 1.0, 64, 32768, False)
    fake = rs["rgb_fine"].reshape(1, PSX, PSY, 3).permute(0, 3, 1, 2)
    np.random.seed(9); torch.manual_seed(9)
    pr, pf_ = Dc(batch["real_patch"]), Dc(fake)
    want = (torch.relu(1 - pr).mean() + torch.relu(1 + pf_).mean()) / 2
    # (the returned 'loss' is what the reference back-propagates in this pass, loss_d * dis_weight -- sinnerf.py:499; the log holds loss_d)
    got_d = out_d["log"]["train/loss_d"].item()
    assert abs(got_d - want.item()) <= 1e-4 * max(1.0, abs(want.item())), (got_d, want.item())
    assert abs(out_d["loss"].item() - sysm.hparams.dis_weight * want.item()) <= 1e-4 * sysm.hparams.dis_weight * max(1.0, abs(want.item()))
    # ... and the driver runs both passes: D moves in pass 1 only, the NeRFs receive nothing from it
    w1 = [p.detach().clone() for p in sysm.nerf_fine.parameters()]
    out_g, out_d2 = sysm.train_step_adversarial(batch)
    assert torch.isfinite(out_g["loss"]).item() and torch.isfinite(out_d2["loss"]).item()
    assert any(not torch.equal(a, b.detach()) for a, b in zip(d0, D.parameters()))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w1, sysm.nerf_fine.parameters()))
