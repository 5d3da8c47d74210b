#!/usr/bin/env python3
"""Generate golden vectors from the UNMODIFIED reference (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--grads | --rays | --loss | --pfm | --classic-heads]

Imports ``models.rendering`` / ``models.nerf`` straight from ``/root/reference`` (read-only mount,
never present on the GPU box), executes them on CPU/fp32 with seeded synthetic inputs and stores
inputs' recipe + outputs as small ``tests/golden/*.npz`` fixtures.  Weights are produced by
``oracle_np.init_params(seed)`` (numpy RandomState: stable across torch versions) and loaded into the
reference's ``NeRF`` through ``load_state_dict`` so that a fixture only needs to carry the seed.
Random draws made by the reference (``torch.rand`` / ``torch.randn``) are recorded in call order and
stored so the oracle / HIP path can be fed the very same numbers.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("SINNERF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)          # reference first: its `datasets` pkg name clashes with HF datasets
sys.path.insert(1, REPO)

from models.nerf import NeRF, Embedding                      # noqa: E402  (reference)
from models import rendering as ref_rendering                # noqa: E402  (reference)
from oracle import oracle_np as O                            # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)


def ref_model(seed, teacher, new_act=True):
    """seed: int -> oracle_np.init_params(seed, teacher); "trained:coarse" / "trained:fine" -> the trained-student fixture."""
    m = NeRF(use_new_activation=new_act)
    p = O.trained_params(seed.split(":")[1]) if isinstance(seed, str) else O.init_params(seed, teacher)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m.eval(), p


class RngTap:
    """Record torch.rand/randn draws made inside the reference, in order."""

    def __enter__(self):
        self.draws = []
        self._rand, self._randn = torch.rand, torch.randn

        def rand(*a, **k):
            t = self._rand(*a, **k); self.draws.append(("rand", t.numpy().copy())); return t

        def randn(*a, **k):
            t = self._randn(*a, **k); self.draws.append(("randn", t.numpy().copy())); return t
        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


def llff_like_rays(n, seed):
    r = np.random.RandomState(seed)
    W, H = 504, 378
    f = W * 0.82
    idx = r.choice(W * H, n, replace=False)
    i, j = (idx % W).astype(np.float64), (idx // W).astype(np.float64)
    d = np.stack([(i - W / 2) / f, -(j - H / 2) / f, -np.ones_like(i)], -1)
    o = np.broadcast_to(np.array([0.1, -0.05, 0.2]), d.shape)
    rays = np.concatenate([o, d, np.full((n, 1), 1.2), np.full((n, 1), 8.0)], 1).astype(np.float32)
    return rays


def case_render(name, rays, seeds, teacher, **kw):
    mc, _ = ref_model(seeds[0], teacher)
    mf, _ = ref_model(seeds[1], teacher)
    emb = [Embedding(3, 10), Embedding(3, 4)]
    torch.manual_seed(1234)
    with torch.no_grad(), RngTap() as tap:
        res = ref_rendering.render_rays([mc, mf], emb, torch.from_numpy(rays),
                                        kw["N_samples"], kw.get("use_disp", False), kw["perturb"], kw["noise_std"],
                                        kw["N_importance"], kw.get("chunk", 32768), kw["white_back"],
                                        test_time=kw.get("test_time", False))
    rng = {}
    kinds = [k for k, _ in tap.draws]
    draws = [d for _, d in tap.draws]
    # consumption order: [perturb rand] -> coarse noise randn -> [u rand] -> fine noise randn
    it = iter(zip(kinds, draws))
    if kw["perturb"] > 0:
        k, d = next(it); assert k == "rand"; rng["perturb"] = d
    k, d = next(it); assert k == "randn"; rng["noise_coarse"] = d
    if kw["N_importance"] > 0:
        if kw["perturb"] > 0:
            k, d = next(it); assert k == "rand"; rng["u"] = d
        k, d = next(it); assert k == "randn"; rng["noise_fine"] = d
    assert next(it, None) is None
    trained = isinstance(seeds[0], str)
    meta = dict(seed_coarse=-1 if trained else seeds[0], seed_fine=-1 if trained else seeds[1], teacher=int(teacher),
                weights="trained_student" if trained else "init",
                use_disp=int(kw.get("use_disp", False)), test_time=int(kw.get("test_time", False)),
                chunk=kw.get("chunk", 32768),
                **{k: kw[k] for k in ("N_samples", "perturb", "noise_std", "N_importance", "white_back")})
    arrays = {"rays": rays}
    arrays.update({"meta_" + k: np.asarray(v) for k, v in meta.items()})
    arrays.update({"rng_" + k: v for k, v in rng.items()})
    arrays.update({"out_" + k: v.numpy() for k, v in res.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name, {k: tuple(v.shape) for k, v in res.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    lego = O.lego_rays(400, 400, seed=0)
    sel = np.random.RandomState(7).choice(lego.shape[0], 192, replace=False)
    lego_s = np.ascontiguousarray(lego[sel])

    case_render("render_lego_eval_teacher", lego_s, (0, 1), True, N_samples=64, perturb=0, noise_std=0,
                N_importance=64, white_back=True)
    case_render("render_lego_eval_rawinit", lego_s[:96], (2, 3), False, N_samples=64, perturb=0, noise_std=0,
                N_importance=64, white_back=True, chunk=1024)
    case_render("render_llff_eval_128", llff_like_rays(96, 5), (0, 1), True, N_samples=64, perturb=0, noise_std=0,
                N_importance=128, white_back=False)
    case_render("render_lego_train_teacher", lego_s[:128], (0, 1), True, N_samples=64, perturb=1.0, noise_std=1.0,
                N_importance=64, white_back=True)
    case_render("render_lego_testtime", lego_s[:64], (0, 1), True, N_samples=64, perturb=0, noise_std=0,
                N_importance=64, white_back=True, test_time=True)
    case_render("render_llff_disp_coarse_only", llff_like_rays(64, 9), (4, 5), True, N_samples=64, perturb=0.5,
                noise_std=0.0, N_importance=0, white_back=False, use_disp=True)
    case_render("render_ragged_small", lego_s[:37], (0, 1), True, N_samples=24, perturb=1.0, noise_std=0.5,
                N_importance=40, white_back=False)

    # ---- sample_pdf alone, incl. zero-weight rows and u at the ends (rendering.py:15-61)
    r = np.random.RandomState(11)
    n, m, k = 64, 62, 64
    bins = np.sort(r.uniform(2, 6, (n, m + 1)).astype(np.float32), -1)
    w = r.uniform(0, 1, (n, m)).astype(np.float32) ** 4
    w[0] = 0.0                              # all-zero row -> uniform pdf
    w[1, :30] = 0.0                         # long empty prefix
    w[2] = 0.0; w[2, 17] = 1.0              # delta
    u = r.uniform(0, 1, (n, k)).astype(np.float32)
    u[3, 0] = 0.0; u[3, 1] = 1.0 - 2 ** -24
    det = ref_rendering.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), k, det=True).numpy()
    _rand = torch.rand
    torch.rand = lambda *a, **kk: torch.from_numpy(u.copy())
    try:
        rnd = ref_rendering.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), k, det=False).numpy()
    finally:
        torch.rand = _rand
    np.savez_compressed(os.path.join(OUT, "sample_pdf.npz"), bins=bins, weights=w, u=u, out_det=det, out_rand=rnd)

    # ---- NeRF MLP + Embedding alone (nerf.py)
    m0, _ = ref_model(6, True)
    x3 = r.uniform(-4, 4, (300, 3)).astype(np.float32)
    d3 = r.uniform(-1.2, 1.2, (300, 3)).astype(np.float32)
    with torch.no_grad():
        e_xyz = Embedding(3, 10)(torch.from_numpy(x3))
        e_dir = Embedding(3, 4)(torch.from_numpy(d3))
        full = m0(torch.cat([e_xyz, e_dir], 1))
        sig = m0(e_xyz, sigma_only=True)
    np.savez_compressed(os.path.join(OUT, "nerf_mlp.npz"), seed=np.asarray(6), teacher=np.asarray(1), xyz=x3, dir=d3,
                        emb_xyz=e_xyz.numpy(), emb_dir=e_dir.numpy(), out_full=full.numpy(), out_sigma=sig.numpy())
    # ---- parameter checksums guard init_params() against silent change
    cs = {f"seed{s}_{int(t)}": np.float64(sum(float(np.sum(v, dtype=np.float64)) for v in O.init_params(s, t).values()))
          for s in range(7) for t in (False, True)}
    np.savez_compressed(os.path.join(OUT, "param_checksums.npz"), **cs)
    print("done ->", OUT)


if __name__ == "__main__" and not any(f in sys.argv for f in ("--grads", "--rays", "--loss", "--pfm", "--classic-heads", "--trained")):
    main()


def case_grad(name, rays, seeds, **kw):
    """Golden parameter gradients from the reference's autograd (train mode) for a linear loss with seeded
    coefficients; stores per-tensor norms, 256 sampled entries per tensor and the full bias gradients."""
    mc, _ = ref_model(seeds[0], True)
    mf, _ = ref_model(seeds[1], True)
    mc.train(); mf.train()
    emb = [Embedding(3, 10), Embedding(3, 4)]
    n = rays.shape[0]
    r = np.random.RandomState(99)
    coef = {"rgb_coarse": r.standard_normal((n, 3)), "depth_coarse": r.standard_normal(n) * 0.3,
            "rgb_fine": r.standard_normal((n, 3)), "depth_fine": r.standard_normal(n) * 0.3}
    coef = {k: v.astype(np.float32) for k, v in coef.items()}
    torch.manual_seed(4321)
    with RngTap() as tap:
        res = ref_rendering.render_rays([mc, mf], emb, torch.from_numpy(rays), kw["N_samples"], False, kw["perturb"],
                                        kw["noise_std"], kw["N_importance"], 32768, kw["white_back"])
        loss = sum((res[k] * torch.from_numpy(v)).sum() for k, v in coef.items())
    loss.backward()
    draws = [d for _, d in tap.draws]
    names = ["perturb", "noise_coarse", "u", "noise_fine"] if kw["perturb"] > 0 else ["noise_coarse", "noise_fine"]
    arrays = {"rays": rays, "loss": np.asarray(loss.item())}
    arrays.update({"rng_" + k: v for k, v in zip(names, draws)})
    arrays.update({"coef_" + k: v for k, v in coef.items()})
    trained = isinstance(seeds[0], str)
    arrays.update({"meta_" + k: np.asarray(v) for k, v in dict(seed_coarse=-1 if trained else seeds[0], seed_fine=-1 if trained else seeds[1],
                                                               weights="trained_student" if trained else "init", **kw).items()})
    arrays.update({"out_" + k: v.detach().numpy() for k, v in res.items()})
    idx_r = np.random.RandomState(5)
    for tag, m in (("coarse", mc), ("fine", mf)):
        for k, p in m.named_parameters():
            g = p.grad.numpy()
            arrays[f"gnorm_{tag}.{k}"] = np.asarray(np.linalg.norm(g.astype(np.float64)))
            if g.ndim == 1:
                arrays[f"gfull_{tag}.{k}"] = g
            else:
                idx = idx_r.choice(g.size, min(256, g.size), replace=False)
                arrays[f"gidx_{tag}.{k}"] = idx
                arrays[f"gval_{tag}.{k}"] = g.reshape(-1)[idx]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name, "loss", loss.item())


def main_grads():
    lego = O.lego_rays(400, 400, seed=0)
    sel = np.random.RandomState(17).choice(lego.shape[0], 96, replace=False)
    case_grad("grad_lego_train", np.ascontiguousarray(lego[sel]), (0, 1), N_samples=64, perturb=1.0, noise_std=1.0,
              N_importance=64, white_back=True)
    case_grad("grad_lego_det", np.ascontiguousarray(lego[sel[:48]]), (2, 3), N_samples=64, perturb=0, noise_std=0,
              N_importance=64, white_back=False)


if __name__ == "__main__" and "--grads" in sys.argv:
    main_grads()


def main_classic_heads():
    """NeRF(use_new_activation=False) (nerf.py:91-100: ReLU / Sigmoid heads, the constructor's default): MLP outputs and the
    reference's own autograd gradients on one batch of embedded points, plus one eval render."""
    os.makedirs(OUT, exist_ok=True)
    m, p = ref_model(6, True, new_act=False)
    r = np.random.RandomState(21)
    n = 160
    xyz = r.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    d = r.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    exyz, edir = Embedding(3, 10), Embedding(3, 4)
    with torch.no_grad():
        x = torch.cat([exyz(torch.from_numpy(xyz)), edir(torch.from_numpy(d))], 1)
    m.train()
    out = m(x)
    g = r.standard_normal((n, 4)).astype(np.float32)
    m.zero_grad()
    out.backward(torch.from_numpy(g))
    grads = {}                                   # as case_grad: norms, 256 sampled entries per matrix, the biases in full
    idx_r = np.random.RandomState(5)
    for k, v in m.named_parameters():
        gr = v.grad.numpy()
        grads["gnorm." + k] = np.asarray(np.linalg.norm(gr.astype(np.float64)))
        if gr.ndim == 1:
            grads["gfull." + k] = gr.copy()
        else:
            idx = idx_r.choice(gr.size, min(256, gr.size), replace=False)
            grads["gidx." + k] = idx
            grads["gval." + k] = gr.reshape(-1)[idx]
    with torch.no_grad():
        sig = m(x[:, :63], sigma_only=True)
    np.savez_compressed(os.path.join(OUT, "nerf_mlp_classic_heads.npz"), seed=6, teacher=True, x=x.numpy(),
                        out=out.detach().numpy(), sigma_only=sig.numpy(), g=g, **grads)
    # one eval render through the reference's render_rays with both networks on the classic heads
    lego = O.lego_rays(400, 400, seed=0)
    rays = np.ascontiguousarray(lego[np.random.RandomState(8).choice(lego.shape[0], 80, replace=False)])
    models = [ref_model(6, True, new_act=False)[0], ref_model(7, True, new_act=False)[0]]
    with torch.no_grad():
        res = ref_rendering.render_rays(models, [exyz, edir], torch.from_numpy(rays), 64, False, 0, 0, 64, 1024 * 32, True)
    np.savez_compressed(os.path.join(OUT, "render_lego_eval_classic_heads.npz"), rays=rays, seeds=np.array([6, 7]),
                        teacher=True, **{k: v.numpy() for k, v in res.items()})
    print("classic heads: out", out.shape, "render keys", sorted(res))


if __name__ == "__main__" and "--classic-heads" in sys.argv:
    main_classic_heads()


def main_rays():
    """datasets/ray_utils.py get_ray_directions / get_rays, run unmodified.  Its only missing import is
    kornia.create_meshgrid (not installed): a stub with kornia's documented semantics -- (1, H, W, 2) pixel grid,
    [..., 0] = x in [0, W-1], [..., 1] = y in [0, H-1] -- is injected for the import."""
    import importlib.util
    import types
    stub = types.ModuleType("kornia")

    def create_meshgrid(height, width, normalized_coordinates=True, **kw):
        assert not normalized_coordinates
        xs, ys = torch.linspace(0, width - 1, width), torch.linspace(0, height - 1, height)
        return torch.stack(torch.meshgrid([xs, ys], indexing="ij"), -1).permute(1, 0, 2).unsqueeze(0)
    stub.create_meshgrid = create_meshgrid
    sys.modules["kornia"] = stub
    spec = importlib.util.spec_from_file_location("ref_ray_utils", os.path.join(REF, "datasets", "ray_utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    H, W, focal, near, far = 21, 34, 41.7, 2.0, 6.0
    r = np.random.RandomState(3)
    q, _ = np.linalg.qr(r.standard_normal((3, 3)))
    c2w = np.concatenate([q, r.uniform(-4, 4, (3, 1))], 1).astype(np.float32)
    directions = ru.get_ray_directions(H, W, focal)
    rays_o, rays_d = ru.get_rays(directions, torch.from_numpy(c2w))
    rays = torch.cat([rays_o, rays_d, near * torch.ones_like(rays_o[:, :1]), far * torch.ones_like(rays_o[:, :1])], 1)
    np.savez_compressed(os.path.join(OUT, "rays.npz"), H=np.asarray(H), W=np.asarray(W), focal=np.asarray(focal),
                        near=np.asarray(near), far=np.asarray(far), c2w=c2w, rays=rays.numpy())
    print("rays", rays.shape)


if __name__ == "__main__" and "--rays" in sys.argv:
    main_rays()


def main_loss():
    """losses.py MSELoss, models/sinnerf.py SL1Loss and metrics.py psnr, run unmodified: the modules' unrelated imports
    that are not installed (torchvision, kornia, piq) are stubbed, and SL1Loss -- whose module imports pytorch_lightning
    & co -- is compiled from its own class node in the reference file (ast), not re-typed."""
    import ast
    import importlib.util
    import types
    for name, attrs in (("torchvision", {"models": None}), ("kornia", {}), ("kornia.losses", {"ssim_loss": None, "ssim": None}),
                        ("piq", {})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.modules["kornia"].losses = sys.modules["kornia.losses"]

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_losses, ref_metrics = load("losses.py", "ref_losses"), load("metrics.py", "ref_metrics")
    tree = ast.parse(open(os.path.join(REF, "models", "sinnerf.py")).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SL1Loss")
    ns = {"nn": torch.nn, "torch": torch}
    exec(compile(ast.Module([node], []), "models/sinnerf.py", "exec"), ns)
    SL1Loss = ns["SL1Loss"]

    r = np.random.RandomState(11)
    n = 777
    t = lambda a: torch.from_numpy(a.astype(np.float32)).requires_grad_(True)
    rgb_gt = r.uniform(0, 1, (n, 3)).astype(np.float32)
    depth_gt = r.uniform(-1.0, 6.0, n).astype(np.float32)            # some <= 0: exercised by useMask=True
    depth_gt[::13] = 0.0
    res = {"rgb_coarse": t(rgb_gt + r.normal(0, 0.2, (n, 3))), "rgb_fine": t(rgb_gt + r.normal(0, 0.05, (n, 3))),
           "depth_coarse": t(depth_gt + r.normal(0, 1.5, n)), "depth_fine": t(depth_gt + r.normal(0, 0.6, n))}
    mask = r.uniform(0, 1, n) < 0.4
    out = {"rgb_gt": rgb_gt, "depth_gt": depth_gt, "mask": mask, **{k: v.detach().numpy() for k, v in res.items()}}
    mse, s1 = ref_losses.MSELoss(), SL1Loss()
    w_depth = 0.37
    for tag, kw in (("nomask", dict(useMask=False)), ("gt0", dict(useMask=True)), ("mask", dict(mask=torch.from_numpy(mask)))):
        for v in res.values():
            v.grad = None
        l2 = mse(res, torch.from_numpy(rgb_gt))["tot"]
        sl_f = s1(res["depth_fine"], torch.from_numpy(depth_gt), **kw)
        sl_c = s1(res["depth_coarse"], torch.from_numpy(depth_gt), **kw)
        total = l2 + w_depth * (sl_f + sl_c)                          # models/sinnerf.py:310-319 shape of the sum
        total.backward()
        out.update({f"{tag}_l2": l2.detach().numpy(), f"{tag}_sl1_fine": sl_f.detach().numpy(),
                    f"{tag}_sl1_coarse": sl_c.detach().numpy(), f"{tag}_total": total.detach().numpy(),
                    **{f"{tag}_g_{k}": v.grad.numpy().copy() for k, v in res.items()}})
    out["psnr_fine"] = ref_metrics.psnr(res["rgb_fine"].detach(), torch.from_numpy(rgb_gt)).numpy()
    out["psnr_coarse"] = ref_metrics.psnr(res["rgb_coarse"].detach(), torch.from_numpy(rgb_gt)).numpy()
    out["mse_fine"] = ref_metrics.mse(res["rgb_fine"].detach(), torch.from_numpy(rgb_gt)).numpy()
    out["w_depth"] = np.asarray(w_depth, np.float32)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)
    print("loss", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__" and "--loss" in sys.argv:
    main_loss()


def main_pfm():
    """datasets/depth_utils.py save_pfm / read_pfm run unmodified (numpy only): the bytes the reference writes."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("ref_depth_utils", os.path.join(REF, "datasets", "depth_utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    r = np.random.RandomState(5)
    grey = r.uniform(0, 6, (7, 5)).astype(np.float32)
    color = r.uniform(0, 1, (4, 6, 3)).astype(np.float32)
    out = {"grey": grey, "color": color}
    with tempfile.TemporaryDirectory() as d:
        for name, img, scale in (("grey", grey, 1), ("color", color, 2.5)):
            path = os.path.join(d, name + ".pfm")
            du.save_pfm(path, img, scale)
            out[name + "_bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8)
            back, sc = du.read_pfm(path)
            out[name + "_read"] = np.ascontiguousarray(back)
            out[name + "_scale"] = np.asarray(sc, np.float32)
    np.savez_compressed(os.path.join(OUT, "pfm.npz"), **out)
    print("pfm", {k: v.shape for k, v in out.items()})


if __name__ == "__main__" and "--pfm" in sys.argv:
    main_pfm()


def main_trained():
    """TRAINED weights: tests/golden/trained_student.npz (tools/train_student.py: the 2 000-step fp32 student of
    tools/convergence.py, trained once on the MI355X) rendered by the unmodified reference -- eval 64+64 lego, 64+128 llff
    (white_back=False), one perturb=1 / noise_std=1 training render, one autograd-gradient case."""
    T = ("trained:coarse", "trained:fine")
    lego = O.lego_rays(400, 400, seed=1)               # the student's held-out pose
    sel = np.random.RandomState(23).choice(lego.shape[0], 192, replace=False)
    lego_s = np.ascontiguousarray(lego[sel])
    case_render("render_trained_lego_eval", lego_s, T, True, N_samples=64, perturb=0, noise_std=0, N_importance=64, white_back=True)
    case_render("render_trained_llff_eval_128", llff_like_rays(96, 15), T, True, N_samples=64, perturb=0, noise_std=0,
                N_importance=128, white_back=False)
    case_render("render_trained_lego_train", lego_s[:128], T, True, N_samples=64, perturb=1.0, noise_std=1.0, N_importance=64,
                white_back=True)
    lego0 = O.lego_rays(400, 400, seed=0)              # a training pose
    sel0 = np.random.RandomState(29).choice(lego0.shape[0], 96, replace=False)
    case_grad("grad_trained_lego_train", np.ascontiguousarray(lego0[sel0]), T, N_samples=64, perturb=1.0, noise_std=1.0,
              N_importance=64, white_back=True)


if __name__ == "__main__" and "--trained" in sys.argv:
    main_trained()
