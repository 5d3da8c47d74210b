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


# developer switch (A/B timing, bit-identity tests): run the compiler-scheduled bf16-state training kernels instead of the
# hand-scheduled ones -- same arithmetic and stored state
COMPILER_SCHEDULED = bool(int(__import__("os").environ.get("SINNERF_COMPILER_SCHEDULED", "0")))
EMB_BF16 = not bool(int(__import__("os").environ.get("SINNERF_EMB_FP32", "0")))    # A/B: keep the fp32 column-order emb


_EMB16_OK = None


def _emb16_supported():
    """the library is asked, once: with SINNERF_DW_NARROW_COMPILER=1 (or a build without the generated narrow kernel) no
    weight-gradient kernel reads a bf16 `emb`, and the forward must not store one"""
    global _EMB16_OK
    if _EMB16_OK is None:
        _EMB16_OK = _lib.lib.sn_weight_grads_workspace_bytes(256, _lib.SN_DTYPE_BF16_STATE | _lib.SN_DTYPE_EMB_BF16) >= 0
    return _EMB16_OK


def _sched_flag():
    return _lib.SN_DTYPE_COMPILER_SCHEDULED if COMPILER_SCHEDULED else 0


def _state_code(model, acts, emb=None):
    """C-ABI dtype of the stored training state: fp32 MFMAs, bf16 operands on fp32 state, or bf16 operands on bf16 state
    (| SN_DTYPE_EMB_BF16 when the embedded inputs were stored as bf16 operands too)."""
    if dtype_code(model.compute_dtype) == _lib.SN_DTYPE_BF16X3:
        return _lib.SN_DTYPE_BF16X3               # 3-term hi/lo splits on the bf16 MFMA; state = (hi, lo) pairs in fp32-sized buffers
    if dtype_code(model.compute_dtype) != _lib.SN_DTYPE_BF16:
        return _lib.SN_DTYPE_F32
    code = _lib.SN_DTYPE_BF16_STATE if acts.dtype == torch.bfloat16 else _lib.SN_DTYPE_BF16
    if emb is not None and emb.dtype == torch.bfloat16:
        code |= _lib.SN_DTYPE_EMB_BF16
    return code


def _sink_of(model, raws, needs):
    """The gradient sink of ``parallel.FlatGradBuffer`` (``param.grad`` = views of one flat buffer), if it is attached to
    every parameter that needs a gradient: the finish kernel then accumulates straight into ``.grad`` (what autograd's
    ``AccumulateGrad`` would do with 24 add launches per network) and backward returns no parameter gradients."""
    sink = getattr(model, "_grad_sink", None)
    if sink is None:
        return None
    for t, s_, need in zip(raws, sink, needs):
        if need and (t.grad is None or t.grad.data_ptr() != s_.data_ptr() or t.grad.shape != t.shape):
            return None
    return sink


def _weight_grads(model, acts, emb, G, needs):
    """dW_l = g_l^T X_l, db_l = sum_p g_l over all sample points (autograd of the nn.Linear layers, nerf.py:66-103):
    ``sn_weight_grads`` = ONE launch of the K-split MFMA kernel for the 14 contractions of the network (plan passed by
    value, nothing built on the host) + one launch that sums the partials in a fixed order and writes the gradients in the
    parameters' shapes.  Returns the list autograd expects (order of ``NeRF.raw_tensors()``)."""
    import ctypes
    dev = acts.device
    code = _state_code(model, acts, emb)
    rows = acts.shape[1]
    nbytes = _lib.lib.sn_weight_grads_workspace_bytes(rows, code)
    if nbytes < 0:
        _lib.check(int(nbytes), "sn_weight_grads_workspace_bytes")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    raws = model.raw_tensors()
    sink = _sink_of(model, raws, needs)
    if sink is not None:
        outs, accumulate = [s_ if need else None for s_, need in zip(sink, needs)], 1
    else:
        flat = torch.empty(sum(t.numel() for t, need in zip(raws, needs) if need), dtype=torch.float32, device=dev)
        outs, off, accumulate = [], 0, 0
        for t, need in zip(raws, needs):
            if need:
                outs.append(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
            else:
                outs.append(None)
    arr = (ctypes.c_void_p * _lib.N_RAW_TENSORS)(*[None if o is None else o.data_ptr() for o in outs])
    _lib.check(_lib.lib.sn_weight_grads(_lib.ptr(acts), _lib.ptr(emb), _lib.ptr(G), rows, code, _lib.ptr(ws), arr, accumulate,
                                        _lib.stream_ptr()), "sn_weight_grads")
    return [None] * len(outs) if sink is not None else outs


class _MLPFn(torch.autograd.Function):
    """rays mode: (rays (N,8), z_vals (N,S)) -> raw (N,S,4).  Embedded mode (``z_vals is None``): ``rays`` is the
    pre-embedded (B, 90) matrix of ``NeRF.forward`` (nerf.py:105-148) -> (B, 4)."""

    @staticmethod
    def forward(ctx, model, rays, z_vals, *params):
        embedded = z_vals is None
        n, s = (rays.shape[0], 1) if embedded else z_vals.shape
        P = n * s
        dev = rays.device
        code = dtype_code(model.compute_dtype)      # (bf16x3: fp32-sized buffers, slots 0..8 written as (hi, lo) bf16 pairs)
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
            _lib.check(_lib.lib.sn_mlp_forward_train_embedded(_lib.ptr(model.packed()), model.kernel_dtype(code), _lib.ptr(rays), n, rays.shape[1],
                                                              _lib.ptr(out), _lib.ptr(acts), rows, _lib.stream_ptr()),
                       "sn_mlp_forward_train_embedded")
        else:
            # NOT zero-filled (that was a 268 MB memset per 4096-ray fine pass): the kernel writes columns [0,63) and [64,91)
            # of EVERY row of the whole point tiles; the pad columns 63, 91..127 only ever feed columns of the 64-wide dW
            # blocks that _weight_grads slices away ([:, :63], [:, :27]) -- a contraction's output column depends on its own
            # X column only
            # bf16 state on the hand-scheduled kernels: emb holds the bf16 operands themselves (SN_DTYPE_EMB_BF16, K-slot order)
            emb16 = bf16 and EMB_BF16 and not COMPILER_SCHEDULED and P < 2 ** 31 - 256 and _emb16_supported()
            emb = torch.empty((rows, 128), dtype=torch.bfloat16 if emb16 else torch.float32, device=dev)
            flags = _sched_flag() | (_lib.SN_DTYPE_EMB_BF16 if emb16 else 0)
            _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(model.packed()), model.kernel_dtype(code) | flags, _lib.ptr(rays), _lib.ptr(z_vals), n, s,
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
            # bf16x3: forward, chain and weight gradients at fp32-level accuracy on the bf16 MFMA; the state holds (hi, lo) pairs
            code = dtype_code(model.compute_dtype)   # bf16: bf16-operand chain on bf16 state; weight gradients and Adam stay fp32
            if acts.dtype == torch.bfloat16:
                code = _lib.SN_DTYPE_BF16_STATE
            _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(model.packed_bwd(model.compute_dtype)), model.kernel_dtype(code) | _sched_flag(),
                                                      _lib.ptr(acts), _lib.ptr(out), _lib.ptr(g_out), P, rows, _lib.ptr(G),
                                                      _lib.ptr(g_o), _lib.stream_ptr()), "sn_mlp_backward_chain")
            needs = ctx.needs_input_grad[3:]
            grads = _weight_grads(model, acts, emb, G, needs)
        return (None, None, None, *grads)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, noise_std, white_back):
        n, s = z_vals.shape
        dev = rays.device
        ctx.set_materialize_grads(False)           # an output no loss term uses arrives as None (the kernel takes a null
        weights = torch.empty((n, s), dtype=torch.float32, device=dev)                 # pointer), not as a zero-filled tensor
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
        if dtype_code(m.compute_dtype) == _lib.SN_DTYPE_F16:       # 'fp32', 'bf16' (mixed precision) and 'bf16x3' have training kernels
            raise NotImplementedError("sinnerf_amd: compute_dtype='fp16' is an INFERENCE arithmetic (sn_mlp_forward only); train in 'bf16' "
                                      "(same matrix rate), 'bf16x3' or 'fp32', or render under torch.no_grad()")
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
    if dtype_code(model.compute_dtype) == _lib.SN_DTYPE_F16:
        raise NotImplementedError("sinnerf_amd: compute_dtype='fp16' is an INFERENCE arithmetic; call under torch.no_grad() or train in "
                                  "'bf16' / 'bf16x3' / 'fp32'")
    if x.requires_grad:
        raise NotImplementedError("sinnerf_amd.NeRF.forward: the gradient with respect to the embedded input x is not "
                                  "implemented (parameter gradients are); detach x, or differentiate through render_rays")
    x = x.contiguous().float()
    if sigma_only:
        x = torch.cat([x, torch.zeros((x.shape[0], 27), dtype=torch.float32, device=x.device)], 1)
    with torch.cuda.device(x.device):
        out = _MLPFn.apply(model, x, None, *model.raw_tensors())
    return out[:, 3:4] if sigma_only else out
