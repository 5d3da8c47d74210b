"""sinnerf_amd -- MI355X-native (gfx950) volume-rendering hot path behind SinNeRF's own Python surface.

Drop-in names (reference: VITA-Group/SinNeRF):
  sinnerf_amd.rendering.render_rays / sample_pdf   <->  models/rendering.py
  sinnerf_amd.nerf.Embedding / NeRF                <->  models/nerf.py
"""
from .nerf import Embedding, NeRF                      # noqa: F401
from .rendering import eval_points, render_rays, sample_pdf   # noqa: F401

__all__ = ["Embedding", "NeRF", "render_rays", "sample_pdf", "eval_points"]
