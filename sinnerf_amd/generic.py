"""The reference's GENERAL configurations on the device, as stock PyTorch-ROCm ops.

The fused HIP kernels implement the one layer configuration both reference call sites build --
``NeRF(D=8, W=256, 63, 27, skips=[4])`` with ``Embedding(3, 10)`` / ``Embedding(3, 4)`` and at most 1024 samples per ray
(``models/sinnerf.py:133-141``, ``eval.py:134-137``).  The reference's constructors are more general
(``models/nerf.py:47-50``: any ``D``, ``W``, ``skips``, input widths; ``nerf.py:8-22``: any ``N_freqs``, linear bands), and a
user who switches over must not lose that: everything outside the fused configuration runs HERE, on the same device, as the
reference's own op sequence (``models/nerf.py:105-148``, ``models/rendering.py:15-61, 126-335``) in torch ops -- differentiable by
autograd, same RNG consumption order, same result dict.  This is a GPU path of the product (eager-PyTorch speed, ~2x slower than
the fused fp32 kernels), not a CPU fallback: CPU tensors still raise, like everywhere else in this package.
"""
import torch
import torch.nn.functional as F


def mlp_generic(model, x, sigma_only=False):
    """``NeRF.forward`` (``models/nerf.py:105-148``) for any layer configuration; activations per ``use_new_activation``
    (``models/activations.py:18-35`` / ``nerf.py:91-100``)."""
    if not sigma_only:
        input_xyz, input_dir = torch.split(x, [model.in_channels_xyz, model.in_channels_dir], dim=-1)   # :123-125
    else:
        input_xyz = x
    h = input_xyz
    for i in range(model.D):                                                   # :131-134
        if i in model.skips:
            h = torch.cat([input_xyz, h], -1)
        lin = getattr(model, f"xyz_encoding_{i+1}")[0]
        h = F.relu(F.linear(h, lin.weight, lin.bias))
    sigma = F.linear(h, model.sigma.weight, model.sigma.bias)                  # :136
    if sigma_only:
        return sigma
    final = F.linear(h, model.xyz_encoding_final.weight, model.xyz_encoding_final.bias)     # :140
    d = F.linear(torch.cat([final, input_dir], -1), model.dir_encoding[0].weight, model.dir_encoding[0].bias)   # :142-143
    if model.use_new_activation:
        d = F.softplus(d - 1.0)                                                # ShiftedSoftplus, activations.py:33-35
        rgb = F.linear(d, model.rgb[0].weight, model.rgb[0].bias)
        rgb = 0.5 * (1.0 + 1.002 * torch.tanh(rgb / 2.0))                      # WidenedSigmoid, activations.py:18-25
    else:
        rgb = torch.sigmoid(F.linear(F.relu(d), model.rgb[0].weight, model.rgb[0].bias))    # nerf.py:91-100
    return torch.cat([rgb, sigma], -1)                                         # :146


def sample_pdf_generic(bins, weights, N_importance, det=False, eps=1e-5):
    """``models/rendering.py:15-61``."""
    n_rays, n_samples = weights.shape
    weights = weights + eps                                                    # :30
    pdf = weights / weights.sum(-1, keepdim=True)                              # :32
    cdf = torch.cumsum(pdf, -1)                                                # :34
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)                   # :36
    if det:
        u = torch.linspace(0, 1, N_importance, device=bins.device).expand(n_rays, N_importance)        # :40-41
    else:
        u = torch.rand(n_rays, N_importance, device=bins.device)              # :43
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)                              # :46
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_samples)
    g = torch.stack([below, above], -1).view(n_rays, 2 * N_importance)         # :50
    cdf_g = torch.gather(cdf, 1, g).view(n_rays, N_importance, 2)
    bins_g = torch.gather(bins, 1, g).view(n_rays, N_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]                                      # :54
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)            # :56 (denom[denom < eps] = 1)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])             # :59-60


def render_generic(models, embeddings, rays, N_samples, use_disp, perturb, noise_std, N_importance, chunk, white_back,
                   test_time, detach_coarse):
    """``models/rendering.py:126-335`` in torch ops (any models / embeddings / sample counts)."""
    embedding_xyz, embedding_dir = embeddings[0], embeddings[1]
    n_rays = rays.shape[0]
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]                                # :257
    near, far = rays[:, 6:7], rays[:, 7:8]                                     # :258
    dir_embedded = embedding_dir(rays_d)                                       # :261

    def inference(model, xyz_, z_vals, weights_only):
        n_samples_ = xyz_.shape[1]
        xyz_ = xyz_.reshape(-1, 3)                                             # :187
        dir_rep = None if weights_only else torch.repeat_interleave(dir_embedded, n_samples_, dim=0)    # :189-190
        outs = []
        for i in range(0, xyz_.shape[0], chunk):                               # :196-204
            xe = embedding_xyz(xyz_[i:i + chunk])
            if not weights_only:
                xe = torch.cat([xe, dir_rep[i:i + chunk]], 1)
            outs.append(mlp_generic(model, xe, sigma_only=weights_only))
        out = torch.cat(outs, 0)                                               # :206
        if weights_only:
            sigmas = out.view(n_rays, n_samples_)
        else:
            rgbsigma = out.view(n_rays, n_samples_, 4)
            rgbs, sigmas = rgbsigma[..., :3], rgbsigma[..., 3]
        deltas = z_vals[:, 1:] - z_vals[:, :-1]                                # :215
        deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)              # :217-218
        deltas = deltas * torch.norm(rays_d.unsqueeze(1), dim=-1)              # :222
        noise = torch.randn(sigmas.shape, device=sigmas.device) * noise_std    # :224
        alphas = 1 - torch.exp(-deltas * torch.relu(sigmas + noise))           # :228
        alphas_shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)  # :229-231
        weights = alphas * torch.cumprod(alphas_shifted, -1)[:, :-1]           # :232-234
        weights_sum = weights.sum(1)                                           # :236
        if weights_only:
            return weights
        rgb_final = torch.sum(weights.unsqueeze(-1) * rgbs, -2)                # :242
        depth_final = torch.sum(weights * z_vals, -1)                          # :243
        if white_back:
            rgb_final = rgb_final + 1 - weights_sum.unsqueeze(-1)              # :246
        return rgb_final, depth_final, weights

    z_steps = torch.linspace(0, 1, N_samples, device=rays.device)             # :264
    if not use_disp:
        z_vals = near * (1 - z_steps) + far * z_steps                          # :268
    else:
        z_vals = 1 / (1 / near * (1 - z_steps) + 1 / far * z_steps)            # :270
    z_vals = z_vals.expand(n_rays, N_samples)
    if perturb > 0:                                                            # :274-282
        z_mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
        upper = torch.cat([z_mid, z_vals[:, -1:]], -1)
        lower = torch.cat([z_vals[:, :1], z_mid], -1)
        z_vals = lower + (upper - lower) * (perturb * torch.rand(z_vals.shape, device=rays.device))
    xyz_coarse = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z_vals.unsqueeze(2)              # :284-285
    result = {}
    if test_time:                                                              # :287-291
        weights_coarse = inference(models[0], xyz_coarse, z_vals, True)
        result["opacity_coarse"] = weights_coarse
    else:
        if detach_coarse:                                                      # :294-298
            with torch.no_grad():
                rgb_coarse, depth_coarse, weights_coarse = inference(models[0], xyz_coarse, z_vals, False)
        else:
            rgb_coarse, depth_coarse, weights_coarse = inference(models[0], xyz_coarse, z_vals, False)
        result.update(rgb_coarse=rgb_coarse, depth_coarse=depth_coarse, opacity_coarse=weights_coarse)
    if N_importance > 0:                                                       # :308-328
        z_mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
        z_new = sample_pdf_generic(z_mid, weights_coarse[:, 1:-1].detach(), N_importance, det=(perturb == 0)).detach()
        z_vals, _ = torch.sort(torch.cat([z_vals, z_new], -1), -1)             # :315
        xyz_fine = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z_vals.unsqueeze(2)
        rgb_fine, depth_fine, weights_fine = inference(models[1], xyz_fine, z_vals, False)
        result.update(rgb_fine=rgb_fine, depth_fine=depth_fine, opacity_fine=weights_fine)
    else:                                                                      # :330-333
        if test_time:
            raise NameError("name 'rgb_coarse' is not defined (render_rays(test_time=True) needs N_importance > 0, "
                            "as in the reference: rendering.py:331)")
        result.update(rgb_fine=result["rgb_coarse"], depth_fine=result["depth_coarse"], opacity_fine=result["opacity_coarse"])
    return result
