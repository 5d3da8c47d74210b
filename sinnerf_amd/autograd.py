"""Training path of the hot path: autograd glue around the backward kernels.

The reference has no explicit backward: PyTorch autograd differentiates ``models/rendering.py`` +
``models/nerf.py`` when Lightning calls ``loss.backward()`` (``models/sinnerf.py:551``).  Here
two ``torch.autograd.Function``s stand in for the two differentiable stages, so the result of ``render_rays`` is
an ordinary autograd graph whose leaves are the ``NeRF`` parameters (DDP hooks / optimisers work unchanged):

  _MLPFn        sn_mlp_forward_train  ->  sn_mlp_backward_chain  ->  dW_l = g_l^T X_l  (big-K GEMMs), db_l = sum g_l
  _CompositeFn  sn_composite_forward  ->  sn_composite_backward

``sample_pdf`` is detached as in the reference (``rendering.py:312``); rays / depths are data (no gradient).
"""
import torch

from . import _lib
from .nerf import dtype_code


def _weight_grads(model, acts, emb, G, g_o, needs):
    """Contractions over all sample points.  Plain GEMMs (hipBLASLt through torch.mm): dW = g^T X, db = sum g.
    Order of the returned list = NeRF.raw_tensors()."""
    grads = []

    def add(gw, gb, k):
        grads.append(gw if needs[2 * k] else None)
        grads.append(gb if needs[2 * k + 1] else None)

    for i in range(8):                                       # xyz_encoding_{i+1}  (nerf.py:66-75)
        gy = G[i]
        if i == 0:
            gw = gy.t() @ emb[:, :63]
        elif i == 4:                                         # skip: cat([input_xyz, h4])  nerf.py:133
            gw = torch.cat([gy.t() @ emb[:, :63], gy.t() @ acts[3]], 1)
        else:
            gw = gy.t() @ acts[i - 1]
        add(gw, gy.sum(0), i)
    h8 = acts[7]
    add(G[8].t() @ h8, G[8].sum(0), 8)                       # xyz_encoding_final  (nerf.py:76)
    gd = G[9][:, :128]
    add(torch.cat([gd.t() @ acts[8], gd.t() @ emb[:, 64:91]], 1), gd.sum(0), 9)       # dir_encoding (nerf.py:142-143)
    gs = g_o[:, 3:4]
    add(gs.t() @ h8, gs.sum(0), 10)                          # sigma (nerf.py:136)
    gr = g_o[:, :3]
    add(gr.t() @ acts[9][:, :128], gr.sum(0), 11)            # rgb (nerf.py:144)
    return grads


class _MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rays, z_vals, *params):
        n, s = z_vals.shape
        P = n * s
        dev = rays.device
        code = dtype_code(model.compute_dtype)
        out = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
        acts = torch.empty((10, P, 256), dtype=torch.float32, device=dev)
        emb = torch.empty((P, 96), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(model.packed()), code, _lib.ptr(rays), _lib.ptr(z_vals), n, s,
                                                 _lib.ptr(out), _lib.ptr(acts), _lib.ptr(emb), _lib.stream_ptr()),
                   "sn_mlp_forward_train")
        ctx.model = model
        ctx.save_for_backward(acts, emb, out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        acts, emb, out = ctx.saved_tensors
        model = ctx.model
        P = acts.shape[1]
        dev = acts.device
        g_out = g_out.contiguous().float()
        G = torch.empty((10, P, 256), dtype=torch.float32, device=dev)
        g_o = torch.empty((P, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(model.packed_bwd()), dtype_code(model.compute_dtype),
                                                      _lib.ptr(acts), _lib.ptr(out), _lib.ptr(g_out), P, _lib.ptr(G),
                                                      _lib.ptr(g_o), _lib.stream_ptr()), "sn_mlp_backward_chain")
            needs = ctx.needs_input_grad[3:]
            grads = _weight_grads(model, acts, emb, G, g_o, needs)
        return (None, None, None, *grads)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, noise_std, white_back):
        n, s = z_vals.shape
        dev = rays.device
        weights = torch.empty((n, s), dtype=torch.float32, device=dev)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((n,), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.sn_composite_forward(_lib.ptr(raw), 1, _lib.ptr(z_vals), _lib.ptr(rays), _lib.ptr(noise),
                                                 float(noise_std), n, s, int(bool(white_back)), _lib.ptr(rgb),
                                                 _lib.ptr(depth), _lib.ptr(weights), _lib.stream_ptr()),
                   "sn_composite_forward")
        ctx.save_for_backward(raw, z_vals, rays, noise if noise is not None else torch.empty(0, device=dev))
        ctx.has_noise = noise is not None
        ctx.noise_std, ctx.white_back = float(noise_std), int(bool(white_back))
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_w):
        raw, z_vals, rays, noise = ctx.saved_tensors
        n, s = z_vals.shape
        g_raw = torch.empty((n, s, 4), dtype=torch.float32, device=raw.device)
        c = lambda t: None if t is None else t.contiguous().float()
        g_rgb, g_depth, g_w = c(g_rgb), c(g_depth), c(g_w)
        with torch.cuda.device(raw.device):
            _lib.check(_lib.lib.sn_composite_backward(_lib.ptr(raw), _lib.ptr(z_vals), _lib.ptr(rays),
                                                      _lib.ptr(noise) if ctx.has_noise else None, ctx.noise_std, n, s,
                                                      ctx.white_back, _lib.ptr(g_rgb), _lib.ptr(g_depth), _lib.ptr(g_w),
                                                      _lib.ptr(g_raw), _lib.stream_ptr()), "sn_composite_backward")
        return g_raw, None, None, None, None, None


def render_rays_autograd(models, rays, N_samples, use_disp, perturb, noise_std, N_importance, white_back, test_time,
                         detach_coarse):
    """Differentiable ``render_rays`` (called by ``rendering.render_rays`` when a model parameter requires grad).
    Same stage order and RNG consumption as ``rendering._forward_core``."""
    from . import rendering as R
    dev = rays.device
    n = rays.shape[0]
    stream = _lib.stream_ptr()
    for m in models:
        if dtype_code(m.compute_dtype) != _lib.SN_DTYPE_F32:
            raise NotImplementedError("sinnerf_amd: the training (autograd) path is fp32 in this revision")
    perturb_rand = torch.rand((n, N_samples), device=dev) if perturb > 0 else None
    z_vals = torch.empty((n, N_samples), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.sn_sample_coarse(_lib.ptr(rays), n, N_samples, int(bool(use_disp)), float(perturb),
                                         _lib.ptr(perturb_rand), _lib.ptr(z_vals), stream), "sn_sample_coarse")
    result = {}
    coarse_grad = (not detach_coarse) and (not test_time) and any(p.requires_grad for p in models[0].parameters())
    if test_time:
        # weights-only coarse pass (rendering.py:287-291); its only consumer is the detached sample_pdf
        with torch.no_grad():
            raw_c = R._mlp(models[0], rays, z_vals, sigma_only=True)
    elif coarse_grad:
        raw_c = _MLPFn.apply(models[0], rays, z_vals, *models[0].raw_tensors())
    else:                                                             # detach_coarse: rendering.py:294-298
        with torch.no_grad():
            raw_c = R._mlp(models[0], rays, z_vals, sigma_only=False)
    noise_c = torch.randn((n, N_samples), device=dev)
    noise_c = noise_c if noise_std != 0 else None
    if test_time:
        with torch.no_grad():
            _, _, w_c = R._composite(raw_c, False, z_vals, rays, noise_c, noise_std, white_back)
        result["opacity_coarse"] = w_c
    else:
        if coarse_grad:
            rgb_c, depth_c, w_c = _CompositeFn.apply(raw_c, z_vals, rays, noise_c, noise_std, white_back)
        else:
            with torch.no_grad():
                rgb_c, depth_c, w_c = R._composite(raw_c, True, z_vals, rays, noise_c, noise_std, white_back)
        result.update(rgb_coarse=rgb_c, depth_coarse=depth_c, opacity_coarse=w_c)
    if N_importance > 0:
        u = torch.rand((n, N_importance), device=dev) if perturb > 0 else None
        z_all = torch.empty((n, N_samples + N_importance), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.sn_sample_pdf(_lib.ptr(z_vals), _lib.ptr(w_c.detach()), _lib.ptr(u), n, N_samples,
                                          N_importance, None, _lib.ptr(z_all), stream), "sn_sample_pdf")
        if any(p.requires_grad for p in models[1].parameters()):
            raw_f = _MLPFn.apply(models[1], rays, z_all, *models[1].raw_tensors())
        else:
            with torch.no_grad():
                raw_f = R._mlp(models[1], rays, z_all, sigma_only=False)
        noise_f = torch.randn((n, N_samples + N_importance), device=dev)
        noise_f = noise_f if noise_std != 0 else None
        if raw_f.requires_grad:
            rgb_f, depth_f, w_f = _CompositeFn.apply(raw_f, z_all, rays, noise_f, noise_std, white_back)
        else:
            with torch.no_grad():
                rgb_f, depth_f, w_f = R._composite(raw_f, True, z_all, rays, noise_f, noise_std, white_back)
        result.update(rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=w_f)
    else:
        if test_time:
            raise NameError("name 'rgb_coarse' is not defined (render_rays(test_time=True) needs N_importance > 0, "
                            "as in the reference: rendering.py:331)")
        result.update(rgb_fine=result["rgb_coarse"], depth_fine=result["depth_coarse"],
                      opacity_fine=result["opacity_coarse"])
    return result


def mlp_embedded_autograd(model, x, sigma_only):
    raise NotImplementedError("sinnerf_amd: NeRF.forward on a pre-embedded matrix is inference-only in this revision; "
                              "gradients flow through sinnerf_amd.render_rays")
