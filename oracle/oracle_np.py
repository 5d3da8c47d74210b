"""CPU oracle for the SinNeRF volume-rendering hot path (TEST INFRASTRUCTURE ONLY).

This file is a numpy/fp32 *restatement* of the reference algorithm
(VITA-Group/SinNeRF: ``models/nerf.py``, ``models/activations.py``,
``models/rendering.py``).  It is the checker the HIP path is compared against.
Nothing in the product package (``sinnerf_amd/``) may import it: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do.

Parity pinning: the reference ships no tests / golden vectors, so this
oracle is pinned against outputs of the *reference itself* executed in the build
container (``oracle/gen_golden.py`` imports ``/root/reference`` and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).

Conventions
-----------
* everything is ``np.float32`` and every elementwise op is rounded separately, in
  the order the reference executes it (torch eager = one rounding per op);
* random draws are *inputs* (``rng`` dict) in the order the reference consumes the
  generator: ``perturb`` (N,S) -> ``noise_coarse`` (N,S) -> ``u`` (N,N_imp) ->
  ``noise_fine`` (N,S_f)   [rendering.py:281, :224, :43, :224];
* ``params`` is a dict keyed exactly like ``NeRF.state_dict()``
  (``xyz_encoding_1.0.weight`` ... ``rgb.0.bias``), values ``np.float32`` arrays.
"""
import os

import numpy as np

F = np.float32


# --------------------------------------------------------------------------- nerf.py
def embedding(x, n_freqs):
    """Positional embedding, reference ``models/nerf.py:36-41`` (logscale bands, :20).

    out = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], each 3 wide.
    """
    x = np.asarray(x, F)
    out = [x]
    for k in range(n_freqs):
        f = F(2.0 ** k)
        fx = (f * x).astype(F)          # exact: power-of-two scale
        out.append(np.sin(fx).astype(F))
        out.append(np.cos(fx).astype(F))
    return np.concatenate(out, -1)


def shifted_softplus(x):
    """``models/activations.py:33-35``: log1p(exp(-|x-1|)) + (x-1)*[x-1>=0]."""
    sx = (x - F(1)).astype(F)
    a = np.abs(sx)
    return (np.log1p(np.exp(-a).astype(F)).astype(F) + sx * (sx >= 0).astype(F)).astype(F)


def widened_sigmoid(x):
    """``models/activations.py:18-20``: .5*(1 + 1.002*tanh(.5 x))."""
    scale = F(1.0 + 2.0 * 1e-3)
    return (F(0.5) * (F(1) + scale * np.tanh((F(0.5) * x).astype(F)).astype(F))).astype(F)


def bf16_round(a):
    """Round fp32 -> bf16 -> fp32 (round to nearest even), the conversion ``v_cvt_pk_bf16_f32`` performs on the MFMA
    operands of the bf16-operand kernels.  Used to build the *bf16-emulated* oracle the reduced-precision path is compared
    with (same operand roundings, fp32 accumulation in another order)."""
    a = np.ascontiguousarray(a, F)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(F)
    return np.where(np.isfinite(a), out, a).astype(F)


_OPERAND_ROUND = None          # None = fp32 reference arithmetic; bf16_round = emulate the bf16-operand MFMA path


class bf16_operands:
    """Context manager: every ``_linear`` inside rounds BOTH operands (activations and weights) to bf16, accumulates in
    fp32 and adds the fp32 bias -- the arithmetic of ``v_mfma_f32_32x32x16_bf16`` in csrc/sn_mlp_fwd_bf16.hip.
    ``heads_fp32=True`` keeps the two narrow heads (sigma, rgb) in fp32 as the kernels do (VALU work on fp32 values)."""

    def __init__(self, heads_fp32=True):
        self.heads_fp32 = heads_fp32

    def __enter__(self):
        global _OPERAND_ROUND, _HEADS_FP32
        self._saved = (_OPERAND_ROUND, _HEADS_FP32)
        _OPERAND_ROUND, _HEADS_FP32 = bf16_round, self.heads_fp32
        return self

    def __exit__(self, *exc):
        global _OPERAND_ROUND, _HEADS_FP32
        _OPERAND_ROUND, _HEADS_FP32 = self._saved


_HEADS_FP32 = True
_SPLIT3 = False


def fp16_round(a):
    """Round fp32 -> fp16 -> fp32 (RNE, overflow -> inf as v_cvt_pk_f16_f32 does without clamp)."""
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(a, F).astype(np.float16).astype(F)


class fp16_operands(bf16_operands):
    """Context manager: as ``bf16_operands`` with fp16-rounded operands (11 significand bits instead of 8; v_mfma_f32_32x32x16_f16
    runs at the bf16 rate on gfx950).  ORACLE-SIDE EXPERIMENT ONLY: no kernel implements it; tools/precision_probe.py
    prints what it would buy on the trained-weight fixtures."""

    def __enter__(self):
        global _OPERAND_ROUND
        super().__enter__()
        _OPERAND_ROUND = fp16_round
        return self


def bf16_split(a):
    """(hi, lo) with hi = RNE_bf16(a), lo = RNE_bf16(a - hi): the operand pair of the 3-term split (16 mantissa bits)."""
    hi = bf16_round(a)
    return hi, bf16_round((np.asarray(a, F) - hi).astype(F))


class bf16x3_operands(bf16_operands):
    """Context manager: every non-head ``_linear`` inside computes  Wh.xh + Wl.xh + Wh.xl  (fp32 accumulate) with the (hi, lo)
    bf16 pairs of both operands -- the arithmetic of csrc/sn_mlp_fwd_bf16x3.hip (SN_DTYPE_BF16X3: fp32-level accuracy on the bf16
    MFMA; the dropped Wl.xl term is 2^-16 relative).  Heads stay fp32 as in the kernel."""

    def __enter__(self):
        global _SPLIT3
        super().__enter__()
        self._saved3, _SPLIT3 = _SPLIT3, True
        return self

    def __exit__(self, *exc):
        global _SPLIT3
        _SPLIT3 = self._saved3
        super().__exit__(*exc)


def _linear(x, w, b, head=False):
    """nn.Linear: x @ W^T + b (``nerf.py:68-76``)."""
    if _OPERAND_ROUND is not None and not (head and _HEADS_FP32):
        if _SPLIT3:
            (xh, xl), (wh, wl) = bf16_split(x), bf16_split(w)
            return ((xh @ wh.T + b) + (xh @ wl.T) + (xl @ wh.T)).astype(F)
        return (_OPERAND_ROUND(x) @ _OPERAND_ROUND(w).T + b).astype(F)
    return (x @ w.T + b).astype(F)


# ``NeRF(use_new_activation=...)`` (nerf.py:47-50): True (both reference call sites) = ShiftedSoftplus / WidenedSigmoid heads,
# False (the constructor's default) = ReLU / Sigmoid.  Switched for a block of oracle calls by ``with classic_heads():``.
_NEW_ACT = True


class classic_heads:
    """``with classic_heads():`` -- nerf_forward / nerf_backward (and everything built on them) restate
    ``NeRF(use_new_activation=False)`` (nerf.py:91-100) inside the block."""

    def __enter__(self):
        global _NEW_ACT
        self._old, _NEW_ACT = _NEW_ACT, False
        return self

    def __exit__(self, *exc):
        global _NEW_ACT
        _NEW_ACT = self._old


def sigmoid(x):
    """nn.Sigmoid in fp32 (nerf.py:100)."""
    x = np.asarray(x, F)
    with np.errstate(over="ignore"):
        return (F(1) / (F(1) + np.exp(-x))).astype(F)


def nerf_forward(params, x, sigma_only=False, D=8, W=256, in_xyz=63, in_dir=27, skips=(4,),
                 return_hidden=False, cache=None):
    """NeRF MLP forward, reference ``models/nerf.py:122-148``; ``use_new_activation=True`` (both call sites:
    sinnerf.py:137,140 / eval.py:136-137) unless called inside ``with classic_heads():``.

    x: (B, 63[+27]) embedded input.  Returns (B,1) raw sigma if sigma_only else (B,4)=[rgb,sigma].
    """
    x = np.asarray(x, F)
    if not sigma_only:
        input_xyz, input_dir = x[:, :in_xyz], x[:, in_xyz:in_xyz + in_dir]   # :123-125
    else:
        input_xyz = x
    h = input_xyz
    hidden = []
    for i in range(D):                                                       # :131-134
        if i in skips:
            h = np.concatenate([input_xyz, h], -1)                           # :133
        h = _linear(h, params[f"xyz_encoding_{i+1}.0.weight"], params[f"xyz_encoding_{i+1}.0.bias"])
        h = np.maximum(h, F(0))                                              # ReLU, :73
        hidden.append(h)
        if cache is not None:
            cache[f"h{i+1}"] = h
    sigma = _linear(h, params["sigma.weight"], params["sigma.bias"], head=True)        # :136
    if sigma_only:
        return sigma
    final = _linear(h, params["xyz_encoding_final.weight"], params["xyz_encoding_final.bias"])  # :140
    d_in = np.concatenate([final, input_dir], -1)                            # :142
    y2 = _linear(d_in, params["dir_encoding.0.weight"], params["dir_encoding.0.bias"])
    d = shifted_softplus(y2) if _NEW_ACT else np.maximum(y2, F(0))          # :143 (ShiftedSoftplus, nerf.py:81-84 / ReLU, :91-94)
    y3 = _linear(d, params["rgb.0.weight"], params["rgb.0.bias"], head=True)
    rgb = widened_sigmoid(y3) if _NEW_ACT else sigmoid(y3)                   # :144 (WidenedSigmoid, nerf.py:86-90 / Sigmoid, :96-100)
    out = np.concatenate([rgb, sigma], -1)                                   # :146
    if cache is not None:
        cache.update(x=x, final=final, y2=y2, d=d, y3=y3, new_act=_NEW_ACT)
    if return_hidden:
        return out, hidden + [final, d]
    return out


# --------------------------------------------------------------------- rendering.py
def linspace01(n):
    """torch.linspace(0, 1, n) in fp32 as the CPU kernel evaluates it: step = 1/(n-1) in fp32;
    lower half i*step, upper half fma(-(n-1-i), step, 1) (probe-verified bit-exact for
    n in {5,7,33,64,128,192}).  Used at ``rendering.py:264`` and ``:40``."""
    if n == 1:
        return np.zeros(1, F)
    step = F(1) / F(n - 1)
    i = np.arange(n)
    lo = (i.astype(F) * step).astype(F)
    hi = (1.0 - (n - 1 - i).astype(np.float64) * np.float64(step)).astype(F)   # single rounding = fma
    return np.where(i < n // 2, lo, hi).astype(F)


def sample_pdf(bins, weights, n_importance, det=False, u=None, eps=1e-5):
    """Inverse-CDF importance sampling, reference ``models/rendering.py:15-61``.

    bins (N, M+1), weights (N, M).  ``u`` (N, n_importance) replaces ``torch.rand`` (:43) when det=False.
    """
    bins = np.asarray(bins, F)
    n_rays, m = weights.shape
    eps32 = F(eps)
    w = (np.asarray(weights, F) + eps32).astype(F)                            # :30
    pdf = (w / np.sum(w, -1, keepdims=True, dtype=F)).astype(F)               # :32
    # torch's CPU cumsum/cumprod accumulate float inputs in double (acc_type) and round once per
    # element (probe: bit-exact against torch 2.10 CPU); plain fp32 sequential scans are not.
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(F)                     # :34
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)                # :36
    if det:
        u = np.broadcast_to(linspace01(n_importance), (n_rays, n_importance)) # :40-41
    else:
        assert u is not None and u.shape == (n_rays, n_importance)
    u = np.ascontiguousarray(u, F)
    # searchsorted(cdf, u, right=True) == number of cdf entries <= u         # :46
    inds = np.empty((n_rays, n_importance), np.int64)
    step = max(1, (1 << 22) // max(1, n_importance * (m + 1)))
    for s in range(0, n_rays, step):
        inds[s:s + step] = (cdf[s:s + step, None, :] <= u[s:s + step, :, None]).sum(-1)
    below = np.maximum(inds - 1, 0)                                           # :47
    above = np.minimum(inds, m)                                               # :48
    cdf_b = np.take_along_axis(cdf, below, 1)                                 # :50-52
    cdf_a = np.take_along_axis(cdf, above, 1)
    bin_b = np.take_along_axis(bins, below, 1)
    bin_a = np.take_along_axis(bins, above, 1)
    denom = (cdf_a - cdf_b).astype(F)                                         # :54
    denom = np.where(denom < eps32, F(1), denom)                              # :56
    samples = (bin_b + ((u - cdf_b).astype(F) / denom).astype(F) * (bin_a - bin_b).astype(F)).astype(F)  # :59-60
    return samples


def composite(rgbsigma, z_vals, rays_d, noise, noise_std, white_back, weights_only=False):
    """Alpha compositing, reference closure ``inference`` ``models/rendering.py:215-248``.

    rgbsigma: (N,S,4) [rgb, raw sigma] or (N,S) raw sigma when weights_only.
    noise: (N,S) standard normal draws (the reference always draws them, :224) or None (=0).
    """
    z_vals = np.asarray(z_vals, F)
    if weights_only:
        sigmas = np.asarray(rgbsigma, F)
    else:
        rgbs = rgbsigma[..., :3]
        sigmas = rgbsigma[..., 3]
    deltas = (z_vals[:, 1:] - z_vals[:, :-1]).astype(F)                       # :215
    deltas = np.concatenate([deltas, np.full_like(deltas[:, :1], 1e10)], -1)  # :217-218
    dnorm = np.sqrt(np.sum((rays_d * rays_d).astype(F), -1, keepdims=True, dtype=F)).astype(F)
    deltas = (deltas * dnorm).astype(F)                                       # :222
    if noise is None:
        nz = np.zeros_like(sigmas)
    else:
        nz = (np.asarray(noise, F) * F(noise_std)).astype(F)                  # :224
    alphas = (F(1) - np.exp((-deltas * np.maximum((sigmas + nz).astype(F), F(0))).astype(F)).astype(F)).astype(F)  # :228
    shifted = np.concatenate([np.ones_like(alphas[:, :1]),
                              ((F(1) - alphas).astype(F) + F(1e-10)).astype(F)], -1)                 # :229-231
    trans = np.cumprod(shifted.astype(np.float64), -1).astype(F)[:, :-1]      # :233 (double accumulate, see sample_pdf)
    weights = (alphas * trans).astype(F)                                      # :232-234
    if weights_only:
        return weights                                                        # :238-239
    wsum = np.sum(weights, 1, dtype=F)                                        # :236
    rgb = np.sum((weights[..., None] * rgbs).astype(F), -2, dtype=F)          # :242
    depth = np.sum((weights * z_vals).astype(F), -1, dtype=F)                 # :243
    if white_back:
        rgb = ((rgb + F(1)).astype(F) - wsum[:, None]).astype(F)              # :246
    return rgb, depth, weights


def coarse_z_vals(rays, n_samples, use_disp=False, perturb=0.0, perturb_rand=None):
    """Stratified depths, reference ``models/rendering.py:264-282``."""
    rays = np.asarray(rays, F)
    n_rays = rays.shape[0]
    near, far = rays[:, 6:7], rays[:, 7:8]                                    # :258
    t = linspace01(n_samples)[None, :]                                        # :264
    if not use_disp:
        z = ((near * (F(1) - t).astype(F)).astype(F) + (far * t).astype(F)).astype(F)        # :268
    else:
        z = (F(1) / (((F(1) / near).astype(F) * (F(1) - t).astype(F)).astype(F)
                     + ((F(1) / far).astype(F) * t).astype(F)).astype(F)).astype(F)            # :270
    z = np.broadcast_to(z, (n_rays, n_samples)).astype(F)                     # :272
    if perturb > 0:                                                           # :274-282
        mid = (F(0.5) * (z[:, :-1] + z[:, 1:]).astype(F)).astype(F)
        upper = np.concatenate([mid, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mid], -1)
        pr = (F(perturb) * np.asarray(perturb_rand, F)).astype(F)
        z = (lower + ((upper - lower).astype(F) * pr).astype(F)).astype(F)
    return z


def _points(rays, z):
    """xyz = o + d*z, ``rendering.py:284-285, 317-318``."""
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    return (o + (d * z[:, :, None]).astype(F)).astype(F)


def _run_model(params, rays, z, dir_emb, weights_only, chunk):
    """Point-chunk loop of closure ``inference`` (``rendering.py:185-212``)."""
    n_rays, s = z.shape
    xyz = _points(rays, z).reshape(-1, 3)                                     # :187
    if not weights_only:
        de = np.repeat(dir_emb, s, axis=0)                                    # :189-190
    outs = []
    for i in range(0, xyz.shape[0], chunk):                                   # :196
        xe = embedding(xyz[i:i + chunk], 10)                                  # :198
        if not weights_only:
            xe = np.concatenate([xe, de[i:i + chunk]], 1)                     # :200-201
        outs.append(nerf_forward(params, xe, sigma_only=weights_only))       # :204
    out = np.concatenate(outs, 0)                                             # :206
    return out.reshape(n_rays, s) if weights_only else out.reshape(n_rays, s, 4)


def render_rays(models, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                chunk=1024 * 32, white_back=False, test_time=False, rng=None, return_z=False):
    """Reference ``models/rendering.py:126-335``.  ``models`` = [coarse_params] or [coarse, fine]
    (state_dict-keyed dicts).  ``rng`` supplies the draws (see module docstring); missing keys = zeros
    (noise) and are required when perturb>0 (``perturb``, ``u``)."""
    rng = rng or {}
    rays = np.asarray(rays, F)
    rays_d = rays[:, 3:6]
    dir_emb = embedding(rays_d, 4)                                            # :261
    z = coarse_z_vals(rays, N_samples, use_disp, perturb, rng.get("perturb"))  # :264-282
    z_coarse = z
    result = {}
    if test_time:                                                             # :287-291
        sig = _run_model(models[0], rays, z, dir_emb, True, chunk)
        w_c = composite(sig, z, rays_d, rng.get("noise_coarse"), noise_std, white_back, weights_only=True)
        result["opacity_coarse"] = w_c
    else:                                                                     # :293-306
        raw = _run_model(models[0], rays, z, dir_emb, False, chunk)
        rgb_c, depth_c, w_c = composite(raw, z, rays_d, rng.get("noise_coarse"), noise_std, white_back)
        result.update(rgb_coarse=rgb_c, depth_coarse=depth_c, opacity_coarse=w_c)
    if N_importance > 0:                                                      # :308-328
        mid = (F(0.5) * (z[:, :-1] + z[:, 1:]).astype(F)).astype(F)           # :310
        z_f = sample_pdf(mid, w_c[:, 1:-1], N_importance, det=(perturb == 0), u=rng.get("u"))   # :311-312
        z = np.sort(np.concatenate([z, z_f], -1), -1)                         # :315
        raw = _run_model(models[1], rays, z, dir_emb, False, chunk)           # :322-324
        rgb_f, depth_f, w_f = composite(raw, z, rays_d, rng.get("noise_fine"), noise_std, white_back)
        result.update(rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=w_f)
    else:                                                                     # :330-333
        result.update(rgb_fine=result["rgb_coarse"], depth_fine=result["depth_coarse"],
                      opacity_fine=result["opacity_coarse"])
    if return_z:
        result["_z_coarse"] = z_coarse
        result["_z_fine"] = z
    return result


def psnr(a, b):
    """``metrics.py:5-15``: -10 log10(mean((a-b)^2))."""
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return float(-10.0 * np.log10(mse))


# ------------------------------------------------------------ backward (autograd restated)
# ---- losses on the rendered rays (the step after the path) -----------------------------------------------------------
def mse_loss(pred, gt):
    """nn.MSELoss(reduction='mean') (losses.py:15,18-20); metrics.py:5-11 `mse`.  Accumulated in double."""
    d = (pred.astype(np.float32) - gt.astype(np.float32)).astype(np.float32)
    return np.float32(np.mean((d * d).astype(np.float64)))


def smooth_l1_loss(pred, gt, mask=None):
    """nn.SmoothL1Loss(reduction='mean'), beta = 1, over pred[mask], gt[mask] (models/sinnerf.py:36-42)."""
    d = np.abs(pred.astype(np.float32) - gt.astype(np.float32)).astype(np.float32)
    v = np.where(d < 1, np.float32(0.5) * d * d, d - np.float32(0.5)).astype(np.float32)
    if mask is not None:
        v = v[mask]
    return np.float32(np.mean(v.astype(np.float64)))


def render_loss(results, rgbs, depths=None, w_rgb=1.0, w_depth=1.0, mask=None, use_mask=False):
    """MSELoss (losses.py:12-22) + w_depth * SL1 of depth_fine and depth_coarse (models/sinnerf.py:310-319) and the
    gradients of the total w.r.t. the rendered tensors.  Returns (stats dict, grads dict)."""
    st, gr = {}, {}
    total = np.float64(0)
    n3 = rgbs.size if rgbs is not None else 0
    for k in ("coarse", "fine"):
        key = "rgb_" + k
        if rgbs is not None and key in results:
            st["mse_" + k] = mse_loss(results[key], rgbs)
            gr[key] = (np.float32(w_rgb * 2.0 / n3) * (results[key] - rgbs)).astype(np.float32)
            total += w_rgb * st["mse_" + k]
            st["psnr_" + k] = np.float32(-10.0 * np.log10(st["mse_" + k]))
    if depths is not None:
        m = mask if mask is not None else ((depths > 0) if use_mask else np.ones(depths.shape, bool))
        cnt = int(m.sum())
        st["n_depth"] = cnt
        for k in ("coarse", "fine"):
            key = "depth_" + k
            if key in results:
                st["sl1_" + k] = smooth_l1_loss(results[key], depths, m)
                d = (results[key] - depths).astype(np.float32)
                g = np.where(np.abs(d) < 1, d, np.sign(d)).astype(np.float32) * np.float32(w_depth / cnt)
                gr[key] = np.where(m, g, np.float32(0)).astype(np.float32)
                total += w_depth * st["sl1_" + k]
    st["total"] = np.float32(total)
    return st, gr


def nerf_backward(params, cache, g_out, in_xyz=63, D=8, skips=(4,), gy_out=None, operand_round=None):
    """Parameter gradients of ``nerf_forward`` (non sigma_only) for upstream ``g_out`` (B,4) -- what torch autograd
    derives from ``models/nerf.py:122-148`` + ``models/activations.py`` (Linear: gW = g^T x, gb = sum g, gx = g W;
    ReLU(inplace): g*[out>0]; ShiftedSoftplus': sigmoid(x-1); WidenedSigmoid': .2505*(1-tanh(.5x)^2)).
    float64 accumulation (reference = fp32 autograd; compare with a norm-wise tolerance).

    ``operand_round`` (e.g. ``bf16_round``): emulate the mixed-precision backward of csrc/sn_mlp_bwd_bf16.hip + sn_dw.hip
    (SN_DTYPE_BF16_STATE) -- every contraction sees BOTH operands rounded (the pre-activation gradients g_y are rounded once,
    when they are packed for the next transposed layer / stored; activations and weights as in the forward), accumulation
    stays wide; the two narrow transposed heads (rgb.0^T, sigma^T) are fp32 VALU work on unrounded values; the softplus
    derivative is taken from the STORED activation, sigmoid(y2-1) = 1 - exp(-d), as the kernel does.

    ``operand_round="bf16x3"``: emulate the 3-TERM SPLIT backward of csrc/sn_mlp_bwd_bf16x3.hip + sn_dw.hip modes 5-7
    (SN_DTYPE_BF16X3): every operand of every contraction -- pre-activation gradients, transposed weights, activations, embedded
    inputs -- enters as its (hi, lo) bf16 pair (``bf16_split`` of the fp32 value) and a product  A.B  is  Ah.Bh + Al.Bh + Ah.Bl
    (the Al.Bl term is dropped, 2^-16 relative); bias gradients are the column sums of hi + lo; heads and the softplus derivative
    as in the bf16 emulation.  ``gy_out`` then receives the pre-activation gradients as the kernel STORES them (hi + lo)."""
    f8 = np.float64
    x3 = isinstance(operand_round, str) and operand_round == "bf16x3"
    if x3:
        def pair(a):
            hi, lo = bf16_split(np.asarray(a, f8).astype(F))
            return hi.astype(f8), lo.astype(f8)
        rd = lambda a: sum(pair(a))                                          # the value a stored (hi, lo) pair decodes to
        def mm(a, b, ta=False):                                              # a (.T) @ b with split operands, wide accumulation
            (ah, al), (bh, bl) = pair(a), pair(b)
            if ta:
                ah, al = ah.T, al.T
            return ah @ bh + al @ bh + ah @ bl
    else:
        rd = (lambda a: operand_round(np.asarray(a, F)).astype(f8)) if operand_round is not None else (lambda a: np.asarray(a, f8))
        mm = lambda a, b, ta=False: (rd(a).T if ta else rd(a)) @ rd(b)
    lowp = operand_round is not None
    g = {}
    x = cache["x"].astype(f8)
    input_xyz, input_dir = x[:, :in_xyz], x[:, in_xyz:]
    g_rgb, g_sigma = g_out[:, :3].astype(f8), g_out[:, 3:4].astype(f8)
    new_act = cache.get("new_act", True)
    if new_act:
        t = np.tanh(0.5 * cache["y3"].astype(f8))
        g_y3 = g_rgb * (0.5 * 1.002 * 0.5) * (1.0 - t * t)
    else:                                                                    # Sigmoid' = s (1 - s)
        sg = 1.0 / (1.0 + np.exp(-cache["y3"].astype(f8)))
        g_y3 = g_rgb * sg * (1.0 - sg)
    d = cache["d"].astype(f8)
    g["rgb.0.weight"], g["rgb.0.bias"] = mm(g_y3, d, ta=True), rd(g_y3).sum(0)
    if gy_out is not None:
        gy_out["rgb"], gy_out["sigma"] = g_y3, g_sigma
    g_d = g_y3 @ params["rgb.0.weight"].astype(f8)
    if not new_act:
        g_y2 = g_d * (d > 0)                                                 # ReLU(inplace): g * [out > 0]
    elif lowp:
        g_y2 = g_d * (1.0 - np.exp(-d))
    else:
        g_y2 = g_d / (1.0 + np.exp(-(cache["y2"].astype(f8) - 1.0)))
    d_in = np.concatenate([cache["final"].astype(f8), input_dir], -1)
    g["dir_encoding.0.weight"], g["dir_encoding.0.bias"] = mm(g_y2, d_in, ta=True), rd(g_y2).sum(0)
    g_final = mm(g_y2, params["dir_encoding.0.weight"])[:, :256]
    if gy_out is not None:
        gy_out["dir"], gy_out["final"] = (g_y2, rd(g_final)) if x3 else (g_y2, g_final)
    h8 = cache[f"h{D}"].astype(f8)
    g["xyz_encoding_final.weight"], g["xyz_encoding_final.bias"] = mm(g_final, h8, ta=True), rd(g_final).sum(0)
    g["sigma.weight"], g["sigma.bias"] = mm(g_sigma, h8, ta=True), rd(g_sigma).sum(0)
    g_h = mm(g_final, params["xyz_encoding_final.weight"]) + g_sigma @ params["sigma.weight"].astype(f8)
    for i in reversed(range(D)):
        h_out = cache[f"h{i+1}"]
        g_y = g_h * (h_out > 0)
        if gy_out is not None:
            gy_out[f"l{i+1}"] = rd(g_y) if x3 else g_y
        if i == 0:
            xin = input_xyz
        else:
            xin = cache[f"h{i}"].astype(f8)
            if i in skips:
                xin = np.concatenate([input_xyz, xin], -1)
        g[f"xyz_encoding_{i+1}.0.weight"], g[f"xyz_encoding_{i+1}.0.bias"] = mm(g_y, xin, ta=True), rd(g_y).sum(0)
        if i > 0:
            g_x = mm(g_y, params[f"xyz_encoding_{i+1}.0.weight"])
            g_h = g_x[:, in_xyz:] if i in skips else g_x
    return g


def composite_backward(rgbsigma, z_vals, rays_d, noise, noise_std, white_back, g_rgb, g_depth, g_w=None):
    """Gradient of ``composite`` w.r.t. ``rgbsigma`` (N,S,4): autograd of ``rendering.py:215-246`` (z_vals / deltas
    carry no gradient: z is data, the fine z_vals are detached at :312).  float64."""
    f8 = np.float64
    z = np.asarray(z_vals, F)
    rgbs, sigmas = rgbsigma[..., :3].astype(f8), rgbsigma[..., 3]
    deltas = (z[:, 1:] - z[:, :-1]).astype(F)
    deltas = np.concatenate([deltas, np.full_like(deltas[:, :1], 1e10)], -1)
    dnorm = np.sqrt(np.sum((rays_d * rays_d).astype(F), -1, keepdims=True, dtype=F)).astype(F)
    deltas = (deltas * dnorm).astype(F).astype(f8)
    nz = np.zeros_like(sigmas) if noise is None else (np.asarray(noise, F) * F(noise_std)).astype(F)
    s_pre = (sigmas + nz).astype(F)
    s = np.maximum(s_pre, F(0)).astype(f8)
    e = np.exp(-deltas * s)
    alphas = 1.0 - e
    f = 1.0 - alphas + 1e-10
    trans = np.cumprod(np.concatenate([np.ones_like(f[:, :1]), f], -1), -1)[:, :-1]
    w = alphas * trans
    G = (g_rgb.astype(f8)[:, None, :] * rgbs).sum(-1) + g_depth.astype(f8)[:, None] * z.astype(f8)
    if g_w is not None:
        G = G + g_w.astype(f8)
    if white_back:
        G = G - g_rgb.astype(f8).sum(-1, keepdims=True)
    Gw = G * w
    suffix = np.flip(np.cumsum(np.flip(Gw, -1), -1), -1) - Gw          # sum_{i>j} G_i w_i
    g_alpha = G * trans - suffix / f
    g_sigma = g_alpha * deltas * e * (s_pre > 0)
    g_raw = np.concatenate([w[..., None] * g_rgb.astype(f8)[:, None, :], g_sigma[..., None]], -1)
    return g_raw


def render_rays_backward(models, rays, upstream, N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                         white_back=False, rng=None, operand_round=None):
    """Parameter gradients of ``render_rays`` for upstream gradients ``upstream`` = dict with any of
    ``rgb_coarse, depth_coarse, opacity_coarse, rgb_fine, depth_fine, opacity_fine``.  Returns [grads_coarse,
    grads_fine] (state_dict-keyed, float64).  ``sample_pdf`` is detached (rendering.py:312)."""
    rng = rng or {}
    rays = np.asarray(rays, F)
    n = rays.shape[0]
    rays_d = rays[:, 3:6]
    dir_emb = embedding(rays_d, 4)
    out = []

    def one(params, z, noise, tag):
        s = z.shape[1]
        xyz = _points(rays, z).reshape(-1, 3)
        xin = np.concatenate([embedding(xyz, 10), np.repeat(dir_emb, s, 0)], 1)
        cache = {}
        raw = nerf_forward(params, xin, cache=cache).reshape(n, s, 4)
        zero3, zero1 = np.zeros((n, 3)), np.zeros((n,))
        g_raw = composite_backward(raw, z, rays_d, noise, noise_std, white_back,
                                   upstream.get("rgb_" + tag, zero3), upstream.get("depth_" + tag, zero1),
                                   upstream.get("opacity_" + tag))
        return raw, nerf_backward(params, cache, g_raw.reshape(-1, 4), operand_round=operand_round)

    z = coarse_z_vals(rays, N_samples, use_disp, perturb, rng.get("perturb"))
    raw_c, g_c = one(models[0], z, rng.get("noise_coarse"), "coarse")
    out.append(g_c)
    if N_importance > 0:
        _, _, w_c = composite(raw_c, z, rays_d, rng.get("noise_coarse"), noise_std, white_back)
        mid = (F(0.5) * (z[:, :-1] + z[:, 1:]).astype(F)).astype(F)
        z_f = sample_pdf(mid, w_c[:, 1:-1], N_importance, det=(perturb == 0), u=rng.get("u"))
        z_all = np.sort(np.concatenate([z, z_f], -1), -1)
        _, g_f = one(models[1], z_all, rng.get("noise_fine"), "fine")
        out.append(g_f)
    return out


def get_rays(H, W, focal, c2w, near, far):
    """``datasets/ray_utils.py:86-133`` (get_ray_directions + get_rays, directions NOT normalised) followed by the
    datasets' ``[rays_o, rays_d, near, far]`` packing (``blender_ray_patch_1image_rot3d.py:201-211``).  (H*W, 8) fp32."""
    c2w = np.asarray(c2w, F)
    j, i = np.meshgrid(np.arange(H, dtype=F), np.arange(W, dtype=F), indexing="ij")
    d = np.stack([((i - F(W / 2)) / F(focal)).astype(F), (-((j - F(H / 2)) / F(focal))).astype(F), -np.ones_like(i)], -1)
    rays_d = (d.reshape(-1, 3) @ c2w[:, :3].T).astype(F)                      # :109
    rays_o = np.broadcast_to(c2w[:, 3], rays_d.shape)                          # :112
    ones = np.ones((H * W, 1), F)
    return np.concatenate([rays_o, rays_d, F(near) * ones, F(far) * ones], 1).astype(F)


# ------------------------------------------------------------------ synthetic inputs
_KEYS = ([f"xyz_encoding_{i+1}.0" for i in range(8)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"])
_SHAPES = {"xyz_encoding_1.0": (256, 63), "xyz_encoding_5.0": (256, 319), "xyz_encoding_final": (256, 256),
           "dir_encoding.0": (128, 283), "sigma": (1, 256), "rgb.0": (3, 128)}


def param_shapes():
    """state_dict key -> shape of ``NeRF(D=8, W=256, 63, 27, skips=[4])`` (``nerf.py:66-103``)."""
    out = {}
    for k in _KEYS:
        shp = _SHAPES.get(k, (256, 256))
        out[k + ".weight"] = shp
        out[k + ".bias"] = (shp[0],)
    return out


def init_params(seed, teacher=False):
    """Seeded stand-in for nn.Linear's default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))), numpy RNG.
    ``teacher=True`` applies SURVEY §8d's non-degenerate-density variant (sigma.weight*8, sigma.bias=.3)."""
    r = np.random.RandomState(seed)
    p = {}
    for k, shp in param_shapes().items():
        fan_in = shp[1] if len(shp) == 2 else param_shapes()[k.replace(".bias", ".weight")][1]
        b = 1.0 / np.sqrt(fan_in)
        p[k] = r.uniform(-b, b, size=shp).astype(F)
    if teacher:
        p["sigma.weight"] = (p["sigma.weight"] * F(8)).astype(F)
        p["sigma.bias"] = np.full((1,), 0.3, F)
    return p


TRAINED_STUDENT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_student.npz")


def trained_params(tag, path=None):
    """TRAINED weights: the 2 000-step fp32 student of tools/convergence.py (tools/train_student.py writes the fixture
    tests/golden/trained_student.npz; tag = "coarse" | "fine"; keys = the reference's ``NeRF.state_dict()``, nerf.py:66-103).
    ``scale`` variants for the precision study live in model_params()."""
    z = np.load(path or TRAINED_STUDENT)
    p = {k[len(tag) + 1:]: np.ascontiguousarray(z[k], F) for k in z.files if k.startswith(tag + ".")}
    assert set(p) == set(param_shapes()), sorted(set(p) ^ set(param_shapes()))
    return p


def model_params(meta):
    """[coarse, fine] parameter dicts of a golden render / gradient case from its ``meta_*`` entries: seeded init weights
    (``seed_coarse`` / ``seed_fine`` / ``teacher``) or, when ``meta["weights"] == "trained_student"``, the trained fixture."""
    if str(meta.get("weights", "")) == "trained_student":
        return [trained_params("coarse"), trained_params("fine")]
    return [init_params(meta["seed_coarse"], bool(meta.get("teacher", True))), init_params(meta["seed_fine"], bool(meta.get("teacher", True)))]


def lego_rays(H, W, seed=0, camera_angle_x=0.6911112, radius=4.0, near=2.0, far=6.0, sel=None):
    """Pin-hole rays as ``datasets/ray_utils.py:86-133`` builds them (un-normalised directions),
    blender/lego intrinsics (``blender_ray_patch_1image_rot3d.py:201-211``), pose on a radius-4 sphere
    looking at the origin (seed-indexed).  Returns (H*W, 8) = [o, d, near, far]."""
    r = np.random.RandomState(1000 + seed)
    focal = 0.5 * 800 / np.tan(0.5 * camera_angle_x) * (W / 800.0)
    th, ph = r.uniform(0, 2 * np.pi), r.uniform(0.15, 0.45) * np.pi
    c = radius * np.array([np.cos(th) * np.sin(ph), np.sin(th) * np.sin(ph), np.cos(ph)])
    fwd = -c / np.linalg.norm(c)
    up = np.array([0, 0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    c2w = np.stack([right, upv, -fwd, c], 1)                                  # (3,4)
    j, i = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    dirs = np.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -np.ones_like(i)], -1)
    rays_d = dirs.reshape(-1, 3) @ c2w[:, :3].T
    rays_o = np.broadcast_to(c2w[:, 3], rays_d.shape)
    rays = np.concatenate([rays_o, rays_d, np.full((H * W, 1), near), np.full((H * W, 1), far)], 1).astype(F)
    if sel is not None:
        rays = rays[sel]
    return np.ascontiguousarray(rays)


def llff_like_rays(n, seed, W=504, H=378, near=1.2, far=8.0):
    """n random pixels of a forward-facing llff/room-shaped frame (504x378, f ~ 0.82 W, near/far ~ 1.2/8,
    ``white_back=False``; the reference's values are data-dependent, ``datasets/llff.py:237-238``)."""
    r = np.random.RandomState(seed)
    f = W * 0.82
    idx = r.choice(W * H, n, replace=False)
    i, j = (idx % W).astype(np.float64), (idx // W).astype(np.float64)
    d = np.stack([(i - W / 2) / f, -(j - H / 2) / f, -np.ones_like(i)], -1)
    o = np.broadcast_to(np.array([0.1, -0.05, 0.2]), d.shape)
    return np.concatenate([o, d, np.full((n, 1), near), np.full((n, 1), far)], 1).astype(F)


def _look_at_c2w(c):
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0, 0, 1.0])); right /= np.linalg.norm(right)
    return np.stack([right, np.cross(right, fwd), -fwd, c], 1)                # (3,4)


def patch_rays(H, W, focal, c2w, near, far, x0, y0, pw, ph, sx, sy):
    """Strided pixel window of a pin-hole frame, row-major over (iy, ix): the patch sampling of the ray-patch datasets
    (``blender_ray_patch_1image_rot3d.py:487-498``, ``llff_ray_patch...:647-668``, ``dtu_proj.py:629-651``) applied to
    ``get_rays``.  Returns (pw*ph, 8)."""
    full = get_rays(H, W, focal, c2w, near, far).reshape(H, W, 8)
    return np.ascontiguousarray(full[y0:y0 + ph * sy:sy, x0:x0 + pw * sx:sx].reshape(-1, 8))


def llff_patch_rays(seed=0, pw=84, ph=63, sx=4, sy=4):
    """BASELINE configs[2]: llff/room 504x378, patch 63x84 (rows x columns) with sW = sH = 4 -> N = 5292 rays, near/far 1.2/8, forward-facing pose."""
    W, H = 504, 378
    r = np.random.RandomState(2000 + seed)
    c2w = np.concatenate([np.eye(3), r.uniform(-0.2, 0.2, (3, 1))], 1)
    return patch_rays(H, W, W * 0.82, c2w, 1.2, 8.0, int(r.randint(0, W - (pw - 1) * sx)), int(r.randint(0, H - (ph - 1) * sy)),
                      pw, ph, sx, sy)


def dtu_patch_rays(seed=0, pw=70, ph=56, sx=8, sy=8):
    """BASELINE configs[3]: dtu scan 640x512, patch 56x70 (rows x columns) with sW = sH = 8 -> N = 3920 rays; near/far 2.125/4.525 (DTU
    depth_min 425 mm, 192 x 2.5 mm intervals, scaled by 1/200: ``dtu_proj.py:290,396-398``), camera ~3.3 units from the
    object looking at it, focal ~ 1.8 W (DTU intrinsics at 640x512), ``white_back=True``."""
    W, H = 640, 512
    r = np.random.RandomState(3000 + seed)
    th, ph_ = r.uniform(0, 2 * np.pi), r.uniform(0.25, 0.4) * np.pi
    c = 3.3 * np.array([np.cos(th) * np.sin(ph_), np.sin(th) * np.sin(ph_), np.cos(ph_)])
    return patch_rays(H, W, 1.8 * W, _look_at_c2w(c), 2.125, 4.525, int(r.randint(0, W - (pw - 1) * sx)),
                      int(r.randint(0, H - (ph - 1) * sy)), pw, ph, sx, sy)
