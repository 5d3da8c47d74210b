"""Plain PyTorch fp32 restatement of the hot path with autograd (TEST INFRASTRUCTURE like the rest of oracle/: only tests/ and
bench.py's baseline legs import it): stock torch ops
only, runs on whatever device its inputs live on.  Used to compare whole optimisation trajectories (SURVEY.md §8d:
"a short seeded training run on both paths") -- the numpy oracle has gradients but no optimiser loop on the GPU box.
Follows models/nerf.py:7-41,46-148, models/activations.py:8-35, models/rendering.py:15-61,126-335."""
import torch
import torch.nn.functional as F


def embed(x, n_freqs):                                         # nerf.py:36-41
    out = [x]
    for k in range(n_freqs):
        out += [torch.sin((2.0 ** k) * x), torch.cos((2.0 ** k) * x)]
    return torch.cat(out, -1)


def nerf(p, x, sigma_only=False):                              # nerf.py:105-148 with use_new_activation=True
    xyz, dirs = x[:, :63], x[:, 63:]
    h = xyz
    for i in range(8):
        if i == 4:
            h = torch.cat([xyz, h], -1)
        h = F.relu(F.linear(h, p[f"xyz_encoding_{i + 1}.0.weight"], p[f"xyz_encoding_{i + 1}.0.bias"]))
    sigma = F.linear(h, p["sigma.weight"], p["sigma.bias"])
    if sigma_only:
        return sigma
    fin = F.linear(h, p["xyz_encoding_final.weight"], p["xyz_encoding_final.bias"])
    d = F.linear(torch.cat([fin, dirs], -1), p["dir_encoding.0.weight"], p["dir_encoding.0.bias"])
    d = F.softplus(d - 1.0)                                    # activations.py:33-35 ShiftedSoftplus
    rgb = F.linear(d, p["rgb.0.weight"], p["rgb.0.bias"])
    rgb = 0.5 * (1.0 + 1.002 * torch.tanh(rgb / 2.0))          # activations.py:18-25 WidenedSigmoid
    return torch.cat([rgb, sigma], -1)


def sample_pdf_det(bins, weights, n_importance, eps=1e-5):      # rendering.py:15-61, det=True
    weights = weights + eps
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0, 1, n_importance, device=bins.device).expand(bins.shape[0], n_importance).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = torch.clamp_min(inds - 1, 0), torch.clamp_max(inds, weights.shape[1])
    g = torch.stack([below, above], -1).view(bins.shape[0], 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, g).view(-1, n_importance, 2)
    bins_g = torch.gather(bins, 1, g).view(-1, n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def render(params, rays, n_samples, n_importance, white_back=True):
    """perturb = 0, noise_std = 0 (deterministic), use_disp = False.  params = [coarse dict, fine dict]."""
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    dir_emb = embed(d, 4)
    t = torch.linspace(0, 1, n_samples, device=rays.device)
    z = near * (1 - t) + far * t

    def inference(p, z):
        n, s = z.shape
        xyz = (o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3)
        raw = nerf(p, torch.cat([embed(xyz, 10), dir_emb.repeat_interleave(s, 0)], 1)).view(n, s, 4)
        deltas = torch.cat([z[:, 1:] - z[:, :-1], 1e10 * torch.ones_like(z[:, :1])], -1) * torch.norm(d[:, None], dim=-1)
        alpha = 1 - torch.exp(-deltas * F.relu(raw[..., 3]))
        trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * trans
        rgb = (w[..., None] * raw[..., :3]).sum(1)
        if white_back:
            rgb = rgb + 1 - w.sum(1, keepdim=True)
        return rgb, (w * z).sum(1), w
    out = {}
    out["rgb_coarse"], out["depth_coarse"], w_c = inference(params[0], z)
    if n_importance > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        z_f = sample_pdf_det(mid, w_c[:, 1:-1].detach(), n_importance).detach()
        z2, _ = torch.sort(torch.cat([z, z_f], -1), -1)
        out["rgb_fine"], out["depth_fine"], _ = inference(params[1], z2)
    return out
