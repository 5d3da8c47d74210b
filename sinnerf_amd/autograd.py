"""Training path (autograd) of the hot path -- reference: implicit autograd of models/rendering.py + nerf.py
(SURVEY.md §8 a10).  Lands after the forward path is parity-green on the GPU."""


def render_rays_autograd(*args, **kwargs):
    raise NotImplementedError("sinnerf_amd: backward kernels (sn_composite_backward / sn_mlp_backward) are not built "
                              "in this revision; call render_rays under torch.no_grad()")


def mlp_embedded_autograd(*args, **kwargs):
    raise NotImplementedError("sinnerf_amd: NeRF.forward backward is not built in this revision; call under "
                              "torch.no_grad()")
