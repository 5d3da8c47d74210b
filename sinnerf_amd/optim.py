"""Flat-buffer Adam -- SURVEY.md §8f rank 3.

Reference: ``utils/__init__.py:11-57`` ``get_optimizer`` -> ``torch.optim.Adam(parameters, lr, eps=1e-8,
weight_decay)`` (``:19-21``), stepped by Lightning after ``backward``.  The two NeRFs have 48 small parameter tensors; a
stock optimiser launches several kernels per tensor.  Here parameters AND gradients of the models live in two flat fp32
buffers (``param.data`` / ``param.grad`` are views), so one step is: one RCCL all-reduce of the flat gradient buffer
(``parallel.FlatGradBuffer``) + ONE ``sn_adam_step`` launch.  State-dict compatible with the reference modules (the views
keep their names and shapes).
"""
import torch

from . import _lib
from .parallel import FlatGradBuffer


class FlatAdam:
    def __init__(self, modules, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._modules = list(modules)
        self.grads = FlatGradBuffer(self._modules)
        params = self.grads.params
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("sinnerf_amd.optim.FlatAdam: parameters must live on a ROCm device (no CPU fallback)")
        self.flat = torch.empty(self.grads.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in params:                              # move every parameter into the flat buffer (keeps values)
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.param_groups = [{"lr": lr}]              # what utils.get_learning_rate / schedulers read (utils/__init__.py:55-57)

    def zero_grad(self, set_to_none=False):
        self.grads.zero()

    def step(self):
        """all-reduce (mean over ranks, no-op at world size 1) + one fused Adam launch."""
        self.grads.all_reduce_mean()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        with torch.cuda.device(self.flat.device):
            _lib.check(_lib.lib.sn_adam_step(_lib.ptr(self.flat), _lib.ptr(self.grads.flat), _lib.ptr(self.exp_avg),
                                             _lib.ptr(self.exp_avg_sq), self.flat.numel(), float(lr), float(self.betas[0]),
                                             float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                             self.step_count, _lib.stream_ptr()), "sn_adam_step")
        # parameters were written behind autograd's back: invalidate the packed-weight caches of the models
        self.generation = getattr(self, "generation", 0) + 1
        for m in self._modules:
            if hasattr(m, "_packed"):
                m._packed.clear()
