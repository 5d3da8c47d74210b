"""Losses on the rendered rays -- the step right after the hot path, same call surface as the
reference's loss objects, computed by the two-launch HIP kernel pair ``sn_render_loss`` (forward value AND the
gradients w.r.t. the rendered tensors in one go; csrc/sn_next.hip):

* ``MSELoss``        losses.py:12-22         ``forward(inputs, targets) -> {'tot', 'l2'}``
* ``SL1Loss``        models/sinnerf.py:32-42 ``forward(depth_pred, depth_gt, mask=None, useMask=True)``
* ``psnr``           metrics.py:14-15
* ``render_loss``    the fused form the system uses: MSE coarse+fine + w_depth * (SL1 coarse + SL1 fine) + PSNR stats

No CPU fallback: CUDA (ROCm) fp32 tensors only.
"""
import torch

from . import _lib

def _workspace(device):
    # ~10 KB of per-block partial sums, allocated per call from torch's caching allocator (stream-ordered: two streams
    # computing losses concurrently never share it)
    return torch.empty(_lib.lib.sn_render_loss_workspace_bytes(), dtype=torch.uint8, device=device)


def _prep(t, n, width, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError(f"{name}: CUDA tensor required (the HIP path has no CPU fallback)")
    t = t.detach().to(torch.float32).reshape(n, width) if width else t.detach().to(torch.float32).reshape(n)
    return t.contiguous()


class _RenderLossFn(torch.autograd.Function):
    """inputs in ABI order; returns (total, stats[8]); stats carries no gradient."""

    @staticmethod
    def forward(ctx, rgb_c, rgb_f, depth_c, depth_f, rgb_gt, depth_gt, mask, mask_mode, w_rgb, w_depth):
        ctx.set_materialize_grads(False)
        preds = (rgb_c, rgb_f, depth_c, depth_f)
        ref = next(t for t in preds if t is not None)
        n = ref.numel() // 3 if ref is rgb_c or ref is rgb_f else ref.numel()
        dev = ref.device
        p = [_prep(rgb_c, n, 3, "rgb_coarse"), _prep(rgb_f, n, 3, "rgb_fine"), _prep(depth_c, n, 0, "depth_coarse"),
             _prep(depth_f, n, 0, "depth_fine")]
        gt_rgb, gt_d = _prep(rgb_gt, n, 3, "rgb_gt"), _prep(depth_gt, n, 0, "depth_gt")
        m = None
        if mask_mode == 2:
            m = mask.detach().reshape(n).to(torch.uint8).contiguous()
        grads = [torch.empty_like(t) if (t is not None and orig.requires_grad) else None
                 for t, orig in zip(p, preds)]
        out = torch.empty(8, dtype=torch.float32, device=dev)
        if n == 0:                                  # empty batch: the reference's 'mean' reductions return NaN (0/0), no raise
            out.fill_(float("nan"))
            out[7] = 0.0
            ctx.shapes = [None if t is None else t.shape for t in preds]
            ctx.save_for_backward(*[g for g in grads if g is not None])
            ctx.have = [g is not None for g in grads]
            ctx.mark_non_differentiable(out)
            return out[4].clone(), out
        with torch.cuda.device(dev):
            ws = _workspace(dev)
            _lib.check(_lib.lib.sn_render_loss(_lib.ptr(p[0]), _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(p[3]),
                                               _lib.ptr(gt_rgb),
                                               _lib.ptr(gt_d), _lib.ptr(m), int(mask_mode), n, float(w_rgb), float(w_depth),
                                               _lib.ptr(grads[0]), _lib.ptr(grads[1]), _lib.ptr(grads[2]), _lib.ptr(grads[3]),
                                               _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()), "sn_render_loss")
        ctx.shapes = [None if t is None else t.shape for t in preds]
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.have = [g is not None for g in grads]
        ctx.mark_non_differentiable(out)
        return out[4].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_stats):
        saved = list(ctx.saved_tensors)
        if g_total is None:
            return (None,) * 10
        scaled = list(torch._foreach_mul(saved, g_total)) if saved else []      # one launch for all four gradients
        res = []
        for have, shape in zip(ctx.have, ctx.shapes):
            res.append(scaled.pop(0).reshape(shape) if have else None)
        return (*res, None, None, None, None, None, None)


def render_loss(results, rgbs=None, depths=None, w_rgb=1.0, w_depth=1.0, mask=None, useMask=False):
    """MSE(rgb_coarse) + MSE(rgb_fine) (losses.py:12-22) + w_depth * (SL1(depth_coarse) + SL1(depth_fine))
    (models/sinnerf.py:310-319, which calls SL1Loss with useMask=False).  Missing keys / targets drop their terms.
    Returns (total, stats) with stats = dict(mse_coarse, mse_fine, sl1_coarse, sl1_fine, psnr_coarse, psnr_fine, n_depth)."""
    mode = 2 if mask is not None else (1 if useMask else 0)
    total, out = _RenderLossFn.apply(results.get("rgb_coarse") if rgbs is not None else None,
                                     results.get("rgb_fine") if rgbs is not None else None,
                                     results.get("depth_coarse") if depths is not None else None,
                                     results.get("depth_fine") if depths is not None else None,
                                     rgbs, depths, mask, mode, w_rgb, w_depth)
    names = ("mse_coarse", "mse_fine", "sl1_coarse", "sl1_fine", "total", "psnr_coarse", "psnr_fine", "n_depth")
    return total, {k: out[i] for i, k in enumerate(names)}


class MSELoss(torch.nn.Module):
    """losses.py:12-22."""

    def forward(self, inputs, targets):
        loss, _ = render_loss(inputs, rgbs=targets)
        return {"tot": loss, "l2": loss}


class SL1Loss(torch.nn.Module):
    """models/sinnerf.py:32-42 (``levels`` is accepted and unused, as there)."""

    def __init__(self, levels=3):
        super().__init__()
        self.levels = levels

    def forward(self, depth_pred, depth_gt, mask=None, useMask=True):
        loss, _ = render_loss({"depth_fine": depth_pred}, depths=depth_gt, mask=mask, useMask=useMask)
        return loss


def psnr(image_pred, image_gt):
    """metrics.py:14-15 (no valid_mask, reduction='mean')."""
    with torch.no_grad():
        _, stats = render_loss({"rgb_fine": image_pred}, rgbs=image_gt)
    return stats["psnr_fine"]
