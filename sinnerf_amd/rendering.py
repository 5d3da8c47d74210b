"""Drop-in ``render_rays`` / ``sample_pdf`` (reference: ``models/rendering.py:15-61, 126-335``).

Same 13-argument signature, same result-dict keys, same consumption order of the global torch RNG
(perturb ``rand`` -> coarse-noise ``randn`` -> ``u`` ``rand`` -> fine-noise ``randn``; rendering.py:281, :224,
:43, :224), so it can be monkey-patched over ``models.rendering.render_rays`` and called unchanged from
``SinNeRF.forward`` (sinnerf.py:177-186) and ``eval.batched_inference`` (eval.py:96-107).

The arithmetic runs in ``libsinnerf_hip.so`` (hand-written HIP for gfx950):

    sn_sample_coarse      rendering.py:264-282   z_vals (+ stratified perturb)
    sn_mlp_forward        rendering.py:187-212 + nerf.py:36-41,122-148   xyz=o+d*z, both embeddings, whole MLP
    sn_composite_forward  rendering.py:215-246   alpha compositing
    sn_sample_pdf         rendering.py:15-61, 310-315   importance sampling + sort

torch is used for allocation, RNG draws and streams only.  There is no CPU fallback: CPU tensors raise.
"""
import torch

from . import _lib
from .nerf import NeRF, dtype_code

__all__ = ["render_rays", "sample_pdf"]

# Optional kernel-level timing hook (bench.py): when set to a list, every MLP launch appends
# (n_points, sigma_only, start_event, end_event) recorded on the launch stream.
PROFILE = None


MAX_FUSED_SAMPLES = 1024       # samples per ray the per-ray kernels (compositor, sampler) hold in one wave


def _is_pow2_bands(e, in_channels, n_freqs):
    """the reference's Embedding keeps no `logscale` attribute (nerf.py:8-22): the frequency bands themselves decide.  The verdict
    is cached on the object, keyed on the band tensor's identity and version: a render call is launch-latency sensitive and must
    not convert 14 band values to Python floats every time (nor sync a device on them if the bands were moved there)."""
    fb = getattr(e, "freq_bands", None)
    if isinstance(fb, torch.Tensor):
        # identity + version + storage: `fb.data = ...` / `set_()` change data_ptr() without bumping _version, a replaced tensor may reuse
        # the id; host bands (10 / 4 values) are cheap enough to key on by VALUE
        key = (id(fb), fb._version, fb.data_ptr(), tuple(fb.shape), str(fb.device), in_channels, n_freqs)
        if fb.device.type == "cpu" and fb.numel() <= 16:
            key += (tuple(fb.detach().reshape(-1).tolist()),)
    else:                                        # list / tuple / ndarray bands: in-place edits are invisible to identity -- key on the values
        try:
            key = (tuple(float(v) for v in fb), in_channels, n_freqs) if fb is not None else (None, in_channels, n_freqs)
        except TypeError:
            key = (id(fb), in_channels, n_freqs)
    hit = getattr(e, "_sn_pow2_verdict", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    ok = (fb is not None and getattr(e, "in_channels", None) == in_channels and getattr(e, "N_freqs", None) == n_freqs
          and len(fb) == n_freqs)
    if ok:
        host = torch.as_tensor(fb).detach().to("cpu", torch.float64)
        ok = bool(torch.equal(host, 2.0 ** torch.arange(n_freqs, dtype=torch.float64)))
    try:
        e._sn_pow2_verdict = (key, ok)
    except Exception:                  # an object that refuses attributes is simply re-checked every call
        pass
    return ok


def _fused_embeddings(embeddings):
    """the kernels fuse Embedding(3, 10) / Embedding(3, 4) with the logscale bands 2^k into the MLP (sinnerf.py:133-134,
    eval.py:134-135); anything else -- including a reference-style Embedding(..., logscale=False) -- is refused"""
    ex, ed = embeddings[0], embeddings[1]
    return _is_pow2_bands(ex, 3, 10) and _is_pow2_bands(ed, 3, 4)


SUPPORTED = ("NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]) with Embedding(3, 10) / Embedding(3, 4) "
             "(logscale bands 2^k) and at most %d samples per ray" % MAX_FUSED_SAMPLES)


def _check_embeddings(embeddings):
    if not _fused_embeddings(embeddings):
        raise NotImplementedError("sinnerf_amd: the HIP kernels fuse the embedding into the MLP and exist for one configuration -- "
                                  + SUPPORTED + " (what models/sinnerf.py:133-141 and eval.py:134-137 build); got other embeddings. "
                                  "There is no torch-op fallback.")


def _mlp(model, rays, z_vals, sigma_only, flags=0):
    """closure ``inference`` of rendering.py:161-212, MLP part: (N,S) depths -> (N,S,4) or (N,S) raw sigma."""
    n, s = z_vals.shape
    out = torch.empty((n, s) if sigma_only else (n, s, 4), dtype=torch.float32, device=rays.device)
    code = dtype_code(model.compute_dtype)
    blob = model.packed()
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(_lib.lib.sn_mlp_forward(_lib.ptr(blob), model.kernel_dtype(code), _lib.ptr(rays), _lib.ptr(z_vals), n, s,
                                       int(sigma_only), flags, _lib.ptr(out), _lib.stream_ptr()), "sn_mlp_forward")
    if PROFILE is not None:
        ev1.record()
        PROFILE.append((n * s, bool(sigma_only), ev0, ev1))
    return out


def _composite(raw, has_rgb, z_vals, rays, noise, noise_std, white_back):
    """closure ``inference`` of rendering.py:214-248, compositing part."""
    n, s = z_vals.shape
    dev = rays.device
    weights = torch.empty((n, s), dtype=torch.float32, device=dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev) if has_rgb else None
    depth = torch.empty((n,), dtype=torch.float32, device=dev) if has_rgb else None
    _lib.check(_lib.lib.sn_composite_forward(_lib.ptr(raw), int(has_rgb), _lib.ptr(z_vals), _lib.ptr(rays),
                                             _lib.ptr(noise), float(noise_std), n, s, int(bool(white_back)),
                                             _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(weights), _lib.stream_ptr()),
               "sn_composite_forward")
    return rgb, depth, weights


def _empty_result(dev, N_samples, N_importance, test_time):
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    sf = N_samples + N_importance
    res = {"opacity_coarse": e(0, N_samples)}
    if not test_time:
        res.update(rgb_coarse=e(0, 3), depth_coarse=e(0))
    elif N_importance == 0:
        raise NameError("name 'rgb_coarse' is not defined (rendering.py:331)")
    res.update(rgb_fine=e(0, 3), depth_fine=e(0), opacity_fine=e(0, sf))
    return res


def _forward_core(models, rays, N_samples, use_disp, perturb, noise_std, N_importance, white_back, test_time,
                  flags=0, keep=None):
    """No-grad forward of the whole path.  ``keep`` (dict) receives the intermediates the backward needs."""
    dev = rays.device
    n = rays.shape[0]
    if n == 0:                                                           # empty batch: the reference returns empty tensors
        return _empty_result(dev, N_samples, N_importance, test_time)
    stream = _lib.stream_ptr()
    perturb_rand = None
    if perturb > 0:                                                      # rendering.py:281
        perturb_rand = torch.rand((n, N_samples), device=dev)
    z_vals = torch.empty((n, N_samples), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.sn_sample_coarse(_lib.ptr(rays), n, N_samples, int(bool(use_disp)), float(perturb),
                                         _lib.ptr(perturb_rand), _lib.ptr(z_vals), stream), "sn_sample_coarse")
    result = {}
    raw_c = _mlp(models[0], rays, z_vals, sigma_only=bool(test_time), flags=flags)
    noise_c = torch.randn((n, N_samples), device=dev)                    # always drawn, rendering.py:224
    rgb_c, depth_c, w_c = _composite(raw_c, not test_time, z_vals, rays, noise_c if noise_std != 0 else None,
                                     noise_std, white_back)
    if test_time:                                                        # rendering.py:287-291
        result["opacity_coarse"] = w_c
    else:                                                                # rendering.py:303-306
        result.update(rgb_coarse=rgb_c, depth_coarse=depth_c, opacity_coarse=w_c)
    if keep is not None:
        keep.update(z_coarse=z_vals, raw_coarse=raw_c, noise_coarse=noise_c if noise_std != 0 else None)
    if N_importance > 0:                                                 # rendering.py:308-328
        u = torch.rand((n, N_importance), device=dev) if perturb > 0 else None      # det = (perturb == 0)
        z_all = torch.empty((n, N_samples + N_importance), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.sn_sample_pdf(_lib.ptr(z_vals), _lib.ptr(w_c), _lib.ptr(u), n, N_samples, N_importance,
                                          None, _lib.ptr(z_all), stream), "sn_sample_pdf")
        raw_f = _mlp(models[1], rays, z_all, sigma_only=False, flags=flags)
        noise_f = torch.randn((n, N_samples + N_importance), device=dev)
        rgb_f, depth_f, w_f = _composite(raw_f, True, z_all, rays, noise_f if noise_std != 0 else None, noise_std,
                                         white_back)
        result.update(rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=w_f)
        if keep is not None:
            keep.update(z_fine=z_all, raw_fine=raw_f, noise_fine=noise_f if noise_std != 0 else None)
    else:                                                                # rendering.py:330-333
        # (the reference raises NameError here when test_time=True; mirrored as KeyError-free explicit error)
        if test_time:
            raise NameError("name 'rgb_coarse' is not defined (render_rays(test_time=True) needs N_importance > 0, "
                            "as in the reference: rendering.py:331)")
        result.update(rgb_fine=rgb_c, depth_fine=depth_c, opacity_fine=w_c)
    return result


def render_rays(models,
                embeddings,
                rays,
                N_samples=64,
                use_disp=False,
                perturb=0,
                noise_std=1,
                N_importance=0,
                chunk=1024 * 32,
                white_back=False,
                test_time=False,
                detach_coarse=False,
                noisy_coarse=True,
                ):
    """Render rays -- drop-in for reference ``models/rendering.py:126-335`` (same arguments, same dict).

    ``chunk`` only bounds temporary memory in the reference (its results are chunk-invariant); the fused kernels
    need no point chunking and ignore it.  ``noisy_coarse`` is unused in the reference as well.  Configurations the kernels
    do not implement (other embeddings, more than 1024 samples per ray; other layer shapes are refused by the ``NeRF``
    constructor) raise ``NotImplementedError``: there is no torch-op second backend.
    """
    if not isinstance(rays, torch.Tensor) or not rays.is_cuda:
        raise RuntimeError("sinnerf_amd.render_rays: rays must be a CUDA/ROCm tensor (there is no CPU fallback)")
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise RuntimeError(f"rays must have shape (N_rays, 8), got {tuple(rays.shape)}")
    for m in models:
        if not isinstance(m, NeRF):
            raise TypeError("sinnerf_amd.render_rays needs sinnerf_amd.NeRF models (state_dict-compatible with "
                            "the reference NeRF; load reference weights with load_state_dict)")
    if N_importance > 0 and len(models) < 2:
        raise IndexError("list index out of range")          # models[1], rendering.py:321
    rays = rays.contiguous().float()
    # one configuration exists in HIP (the one both reference call sites build); everything else is refused, loudly
    _check_embeddings(embeddings)
    if N_samples + N_importance > MAX_FUSED_SAMPLES:
        raise NotImplementedError("sinnerf_amd.render_rays: %d + %d samples per ray; the per-ray kernels (compositor, importance "
                                  "sampler, merge) hold at most %d in one wave.  Supported: %s"
                                  % (N_samples, N_importance, MAX_FUSED_SAMPLES, SUPPORTED))
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for m in models for p in m.parameters())
    with torch.cuda.device(rays.device):
        if needs_grad and rays.shape[0] > 0:
            from .autograd import render_rays_autograd
            return render_rays_autograd(models, rays, N_samples, use_disp, perturb, noise_std, N_importance,
                                        white_back, test_time, detach_coarse)
        with torch.no_grad():
            return _forward_core(models, rays, N_samples, use_disp, perturb, noise_std, N_importance, white_back,
                                 test_time)


@torch.no_grad()
def eval_points(points, models, embeddings):
    """``models/rendering.py:64-123``: raw sigma of the last (fine) model at free points.

    points (B, 3) -> (B, 1).  The reference embeds all points with ``embeddings[0]`` and runs the model in chunks of
    32 768 with ``sigma_only=True``; here the embedding is applied by the same ``Embedding`` module and the fused MLP
    kernel takes the embedded rows in one launch (no chunk loop needed: nothing but the (B, 63) input is materialised).
    (Unused by the reference's own scripts; kept for completeness of the rendering module's surface.)"""
    _check_embeddings(embeddings)
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError(f"eval_points: points must be (B, 3), got {tuple(points.shape)}")
    if points.shape[0] == 0:
        return torch.empty((0, 1), dtype=torch.float32, device=points.device)
    return models[-1](embeddings[0](points.float()), sigma_only=True)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    """Drop-in for reference ``models/rendering.py:15-61``: bins (N, M+1), weights (N, M) -> (N, N_importance)."""
    if not bins.is_cuda:
        raise RuntimeError("sinnerf_amd.sample_pdf: CUDA/ROCm tensors only (no CPU fallback)")
    if not eps > 0:
        raise ValueError("sample_pdf: eps must be positive")
    n, m = weights.shape
    bins = bins.contiguous().float()
    weights = weights.detach().contiguous().float()
    with torch.cuda.device(bins.device):
        u = None if det else torch.rand((n, N_importance), device=bins.device)        # rendering.py:43
        out = torch.empty((n, N_importance), dtype=torch.float32, device=bins.device)
        _lib.check(_lib.lib.sn_sample_pdf_bins(_lib.ptr(bins), _lib.ptr(weights), _lib.ptr(u), n, m, N_importance, float(eps),
                                               _lib.ptr(out), _lib.stream_ptr()), "sn_sample_pdf_bins")
    return out
