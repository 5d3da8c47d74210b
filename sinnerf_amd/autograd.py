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


_VARIANT_MN = {0: (256, 256), 1: (256, 64), 2: (128, 256), 3: (128, 64), 4: (32, 256), 5: (32, 128)}
# cost of one point of a K-range on one CU, in cycles: max(MFMA issue time of the wave block, tile bytes / ~8 B/clk of
# per-CU streaming bandwidth) -- the narrow problems are DMA-bound, not MFMA-bound (measured: splitting by FLOPs alone left
# the 32x128 problem streaming 168 MB through a single CU, 2.5x the kernel time of the balanced split)
_VARIANT_COST = {0: 512, 1: 161, 2: 260, 3: 95, 4: 101, 5: 59}      # measured per-point times (tools/dw_time.py), variant 0 = 512
# bf16-operand mode: 8x less MFMA time, every variant is bound by its per-CU DMA stream -- measured per-point times again
_VARIANT_COST_BF16 = {0: 512, 1: 189, 2: 226, 3: 126, 4: 138, 5: 125}
# ... and with G / the activations stored as bf16 (gather-bound inner loop, half the bytes)
_VARIANT_COST_BF16_STATE = {0: 512, 1: 313, 2: 325, 3: 203, 4: 224, 5: 192}
_KB = 16                      # csrc/sn_dw.hip: points per staged chunk
_TARGET_WGS = 256             # exactly one workgroup per CU per launch


def _dw_tasks(acts, emb, G, bf16=False):
    """Task table of the single sn_dw_gemm launch: the 13 contractions dW = G^T X of a network, K-split over ~one workgroup
    per CU in proportion to their cost.  Returns (rows: list of 8-int64 task records, outs: [(key, partial dW, partial db)])."""
    import numpy as np
    P = acts.shape[1]                                        # padded to a multiple of 16 (pad rows of G are zero)
    dev = acts.device
    # (key, A tensor, A col, lda, B tensor, B col, ldb, variant, want_bias)
    probs = []
    for i in range(8):                                       # xyz_encoding_{i+1}
        if i == 0:
            probs.append((("w", 0), G[0], 0, 256, emb, 0, 128, 1, True))
        else:
            probs.append((("w", i), G[i], 0, 256, acts[i - 1], 0, 256, 0, True))
            if i == 4:                                       # skip: cat([input_xyz, h4])  nerf.py:133
                probs.append((("w4e", 4), G[4], 0, 256, emb, 0, 128, 1, False))
    probs.append((("w", 8), G[8], 0, 256, acts[7], 0, 256, 0, True))          # xyz_encoding_final
    probs.append((("w", 9), G[9], 0, 256, acts[8], 0, 256, 2, True))          # dir_encoding[:, :256]
    probs.append((("w9e", 9), G[9], 0, 256, emb, 64, 128, 3, False))          # dir_encoding[:, 256:]
    # rows 0..2 = g_y of rgb, row 3 = g_y of sigma (zero-padded 32-wide block at G[9][:, 128:160], sn_mlp_bwd.hip)
    probs.append((("sig", 10), G[9], 128, 256, acts[7], 0, 256, 4, False))    # sigma  (nerf.py:136)
    probs.append((("rgb", 11), G[9], 128, 256, acts[9], 0, 256, 5, True))     # rgb    (nerf.py:144)
    cost = (_VARIANT_COST_BF16_STATE if G.dtype == torch.bfloat16 else _VARIANT_COST_BF16) if bf16 else _VARIANT_COST
    # 0x100: bf16 operands; 0x200: G and the activations are STORED as bf16 (emb stays fp32)
    state16 = G.dtype == torch.bfloat16
    assert (not state16) or (bf16 and acts.dtype == torch.bfloat16 and emb.dtype == torch.float32)
    flags = (0x100 if bf16 else 0) | (0x200 if state16 else 0)
    work = [cost[p[7]] for p in probs]
    tot = float(sum(work))
    max_split = max(1, P // (4 * _KB))
    # K-splits proportional to the work of a problem, summing to _TARGET_WGS (largest remainders get the slack)
    ideal = [_TARGET_WGS * w / tot for w in work]
    splits = [max(1, int(x)) for x in ideal]
    for j in sorted(range(len(work)), key=lambda j: ideal[j] - int(ideal[j]), reverse=True):
        if sum(splits) >= _TARGET_WGS:
            break
        splits[j] += 1
    rows, outs = [], []
    for pr, ns in zip(probs, splits):
        key, A, ac, lda, B, bc, ldb, var, want_b = pr
        M, N = _VARIANT_MN[var]
        ns = int(min(max_split, ns))
        per = -(-P // ns)
        per = -(-per // _KB) * _KB
        ns = -(-P // per)
        cpart = torch.empty((ns, M, N), dtype=torch.float32, device=dev)
        bpart = torch.empty((ns, M), dtype=torch.float32, device=dev) if want_b else None
        outs.append((key, cpart, bpart))
        a_ptr, b_ptr = A.data_ptr() + ac * A.element_size(), B.data_ptr() + bc * B.element_size()
        for j in range(ns):
            rows.append((a_ptr, b_ptr, cpart.data_ptr() + j * M * N * 4,
                         (bpart.data_ptr() + j * M * 4) if want_b else 0,
                         j * per, min(P, (j + 1) * per), lda | (ldb << 32), N | ((var | flags) << 32)))
    return rows, outs


def _weight_grads(model, acts, emb, G, g_o, needs):
    """dW_l = g_l^T X_l, db_l = sum_p g_l over all sample points (autograd of the nn.Linear layers, nerf.py:66-103).
    All contractions run in ONE launch of the K-split MFMA kernel (sn_dw_gemm, csrc/sn_dw.hip) followed by a deterministic
    sum of the K-split partials.  Order of the returned list = NeRF.raw_tensors()."""
    import numpy as np
    dev = acts.device
    # mixed precision: bf16-operand contractions (fp32 tiles converted on the fly, fp32 accumulation and partial sums)
    rows, outs = _dw_tasks(acts, emb, G, bf16=dtype_code(model.compute_dtype) == _lib.SN_DTYPE_BF16)
    tasks = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    _lib.check(_lib.lib.sn_dw_gemm(_lib.ptr(tasks), tasks.shape[0], _lib.stream_ptr()), "sn_dw_gemm")
    res = {k: (c.sum(0), b.sum(0) if b is not None else None) for k, c, b in outs}

    grads = []

    def add(gw, gb, k):
        grads.append(gw if needs[2 * k] else None)
        grads.append(gb if needs[2 * k + 1] else None)

    for i in range(8):
        gw, gb = res[("w", i)]
        if i == 0:
            gw = gw[:, :63]
        elif i == 4:
            gw = torch.cat([res[("w4e", 4)][0][:, :63], gw], 1)
        add(gw.contiguous(), gb, i)
    add(*res[("w", 8)], 8)
    gw, gb = res[("w", 9)]
    add(torch.cat([gw, res[("w9e", 9)][0][:, :27]], 1), gb, 9)
    gb4 = res[("rgb", 11)][1]                                # column sums of [g_rgb(3), g_sigma(1), 0...]
    add(res[("sig", 10)][0][3:4].contiguous(), gb4[3:4].contiguous(), 10)
    add(res[("rgb", 11)][0][:3].contiguous(), gb4[:3].contiguous(), 11)
    return grads


class _MLPFn(torch.autograd.Function):
    """rays mode: (rays (N,8), z_vals (N,S)) -> raw (N,S,4).  Embedded mode (``z_vals is None``): ``rays`` is the
    pre-embedded (B, 90) matrix of ``NeRF.forward`` (nerf.py:105-148) -> (B, 4)."""

    @staticmethod
    def forward(ctx, model, rays, z_vals, *params):
        embedded = z_vals is None
        n, s = (rays.shape[0], 1) if embedded else z_vals.shape
        P = n * s
        dev = rays.device
        code = dtype_code(model.compute_dtype)
        # mixed precision keeps the training state (activations, pre-activation gradients) in bf16 as well: every stage of
        # that mode is HBM-bound on exactly this traffic, and the stored values are the ones the kernels consume anyway
        bf16 = code == _lib.SN_DTYPE_BF16
        if bf16:
            code = _lib.SN_DTYPE_BF16_STATE
        out = torch.empty((n, 4) if embedded else (n, s, 4), dtype=torch.float32, device=dev)
        tile = 256 if bf16 else 128
        rows = -(-P // tile) * tile                    # the training forward stores whole point tiles (pad rows: finite
        acts = torch.empty((10, rows, 256), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)   # copies of the last point, zero gradient)
        if embedded:
            # emb = the column layout the weight-gradient contractions read: [0,63) xyz, [64,91) dir (nerf.py:123-125)
            emb = torch.zeros((rows, 128), dtype=torch.float32, device=dev)
            emb[:n, :63] = rays[:, :63]
            emb[:n, 64:91] = rays[:, 63:90]
            _lib.check(_lib.lib.sn_mlp_forward_train_embedded(_lib.ptr(model.packed()), code, _lib.ptr(rays), n, rays.shape[1],
                                                              _lib.ptr(out), _lib.ptr(acts), rows, _lib.stream_ptr()),
                       "sn_mlp_forward_train_embedded")
        else:
            # NOT zero-filled (that was a 268 MB memset per 4096-ray fine pass): the kernel writes columns [0,63) and [64,91)
            # of EVERY row of the whole point tiles; the pad columns 63, 91..127 only ever feed columns of the 64-wide dW
            # blocks that _weight_grads slices away ([:, :63], [:, :27]) -- a contraction's output column depends on its own
            # X column only
            emb = torch.empty((rows, 128), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(model.packed()), code, _lib.ptr(rays), _lib.ptr(z_vals), n, s,
                                                     _lib.ptr(out), _lib.ptr(acts), _lib.ptr(emb), rows, _lib.stream_ptr()),
                       "sn_mlp_forward_train")
        ctx.model = model
        ctx.n_points = P
        ctx.save_for_backward(acts, emb, out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        acts, emb, out = ctx.saved_tensors
        model = ctx.model
        P, rows = ctx.n_points, acts.shape[1]
        dev = acts.device
        g_out = g_out.contiguous().float()               # (N,S,4) or (B,4): P x 4 either way
        G = torch.empty((10, rows, 256), dtype=acts.dtype, device=dev)
        if rows > P:
            G[:, P:].zero_()
        g_o = torch.empty((P, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            code = dtype_code(model.compute_dtype)   # bf16: bf16-operand chain on bf16 state; weight gradients and Adam stay fp32
            if acts.dtype == torch.bfloat16:
                code = _lib.SN_DTYPE_BF16_STATE
            _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(model.packed_bwd(model.compute_dtype)), code,
                                                      _lib.ptr(acts), _lib.ptr(out), _lib.ptr(g_out), P, rows, _lib.ptr(G),
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
        dtype_code(m.compute_dtype)              # 'fp32', or 'bf16' = mixed precision: bf16-operand forward, fp32 backward
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
    """``NeRF.forward(x, sigma_only)`` under autograd (reference ``models/nerf.py:105-148`` is an ordinary differentiable
    module): parameter gradients through the same kernels as ``render_rays`` (training forward on the pre-embedded rows,
    backward chain, weight-gradient contractions).  The gradient w.r.t. ``x`` itself is not implemented (both reference call
    sites feed embeddings of data: rays / sample depths carry no gradient, rendering.py:312) -- asking for it raises.
    ``sigma_only`` (nerf.py:136-138) runs the full network on a zero direction embedding and returns the sigma column;
    the head / dir-branch parameters then receive exact zeros where torch would leave ``.grad`` at None."""
    if x.requires_grad:
        raise NotImplementedError("sinnerf_amd.NeRF.forward: the gradient with respect to the embedded input x is not "
                                  "implemented (parameter gradients are); detach x, or differentiate through render_rays")
    x = x.contiguous().float()
    if sigma_only:
        x = torch.cat([x, torch.zeros((x.shape[0], 27), dtype=torch.float32, device=x.device)], 1)
    with torch.cuda.device(x.device):
        out = _MLPFn.apply(model, x, None, *model.raw_tensors())
    return out[:, 3:4] if sigma_only else out
