"""On-GPU ray generation -- SURVEY.md §8f rank 1, the step immediately before the hot path.

Reference: ``datasets/ray_utils.py:86-133`` (``get_ray_directions`` + ``get_rays``; directions are NOT normalised,
``:110`` is commented out) and the ``[rays_o, rays_d, near, far]`` packing the datasets do (blender:
``blender_ray_patch_1image_rot3d.py:201-211``; strided patches ``:487-498``).  The reference builds the (H*W, 8) array on
the CPU in the DataLoader and copies it to the device (``eval.py:155``); here ``sn_generate_rays`` writes it directly in
HBM from the 12 floats of ``c2w``.
"""
import torch

from . import _lib


def get_rays(H, W, focal, c2w, near, far, window=None):
    """(n, 8) fp32 rays ``[o(3), d(3), near, far]`` on ``c2w``'s device, row-major over pixels.

    ``c2w``: (3,4) camera-to-world tensor on the ROCm device.  ``window = (x0, y0, stride_x, stride_y, patch_w, patch_h)``
    selects a strided patch; default = the full frame."""
    if not c2w.is_cuda:
        raise RuntimeError("sinnerf_amd.ray_utils.get_rays: c2w must be a CUDA/ROCm tensor (no CPU fallback)")
    c2w = c2w.contiguous().float()
    if tuple(c2w.shape) != (3, 4):
        raise RuntimeError(f"c2w must be (3, 4), got {tuple(c2w.shape)}")
    x0, y0, sx, sy, pw, ph = window if window is not None else (0, 0, 1, 1, W, H)
    rays = torch.empty((pw * ph, 8), dtype=torch.float32, device=c2w.device)
    with torch.cuda.device(c2w.device):
        _lib.check(_lib.lib.sn_generate_rays(_lib.ptr(c2w), H, W, float(focal), float(near), float(far), x0, y0, sx, sy, pw,
                                             ph, _lib.ptr(rays), _lib.stream_ptr()), "sn_generate_rays")
    return rays
