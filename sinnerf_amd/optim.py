"""Flat-buffer Adam -- SURVEY.md §8f rank 3.

Reference: ``utils/__init__.py:11-57`` ``get_optimizer`` -> ``torch.optim.Adam(parameters, lr, eps=1e-8,
weight_decay)`` (``:19-21``), stepped by Lightning after ``backward`` and driven by ``MultiStepLR`` (``:27-31``).  The two
NeRFs have 48 small parameter tensors; a stock optimiser launches several kernels per tensor.  Here parameters AND gradients
of the models live in two flat fp32 buffers (``param.data`` / ``param.grad`` are views), so one step is: one RCCL all-reduce
of the flat gradient buffer (``parallel.FlatGradBuffer``) + ONE ``sn_adam_step`` launch.  State-dict compatible with the
reference modules (the views keep their names and shapes).

``FlatAdam`` is a ``torch.optim.Optimizer``: torch LR schedulers (the reference's ``MultiStepLR``) drive
``param_groups[0]['lr']``, ``state_dict()`` / ``load_state_dict()`` carry ``exp_avg`` / ``exp_avg_sq`` / ``step`` for
checkpoint-resume, ``zero_grad()`` keeps the gradient views attached.
"""
import torch

from . import _lib
from .parallel import FlatGradBuffer


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, modules, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._modules = list(modules)
        self.grads = FlatGradBuffer(self._modules)
        params = self.grads.params
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("sinnerf_amd.optim.FlatAdam: parameters must live on a ROCm device (no CPU fallback)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.flat = torch.empty(self.grads.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in params:                              # move every parameter into the flat buffer (keeps values)
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self._invalidate()

    # (the hyper-parameters live in param_groups[0], where schedulers and utils.get_learning_rate -- utils/__init__.py:55-57 --
    #  read and write them)
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def _invalidate(self):
        # parameters were (re)written behind autograd's back: their MFMA-packed copies are stale (blobs are kept and refilled)
        for m in self._modules:
            if hasattr(m, "invalidate_packed"):
                m.invalidate_packed()

    def zero_grad(self, set_to_none=False):
        """One memset of the flat gradient buffer; the ``param.grad`` views stay attached (``set_to_none`` is ignored on
        purpose: detaching the views is what would make the flat buffer go stale)."""
        self.grads.zero()

    @torch.no_grad()
    def step(self, closure=None):
        """all-reduce (mean over ranks, no-op at world size 1) + one fused Adam launch.  Gradients that are no longer views
        of the flat buffer (someone called ``module.zero_grad(set_to_none=True)``) are copied back first
        (``FlatGradBuffer.sync_views``) -- the step never runs on a stale buffer."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.grads.all_reduce_mean()
        self.step_count += 1
        g = self.param_groups[0]
        with torch.cuda.device(self.flat.device):
            _lib.check(_lib.lib.sn_adam_step(_lib.ptr(self.flat), _lib.ptr(self.grads.flat), _lib.ptr(self.exp_avg),
                                             _lib.ptr(self.exp_avg_sq), self.flat.numel(), float(g["lr"]),
                                             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                             float(g["weight_decay"]), self.step_count, _lib.stream_ptr()), "sn_adam_step")
        self._invalidate()
        return loss

    # ---- checkpoint / resume ---------------------------------------------------------------------------------------
    def state_dict(self):
        g = self.param_groups[0]
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"}]}

    def load_state_dict(self, state):
        if state["exp_avg"].numel() != self.flat.numel():
            raise ValueError("FlatAdam.load_state_dict: optimizer state belongs to a different parameter set")
        self.step_count = int(state["step"])
        self.exp_avg.copy_(state["exp_avg"].to(self.flat.device).reshape(-1))
        self.exp_avg_sq.copy_(state["exp_avg_sq"].to(self.flat.device).reshape(-1))
        for k, v in state["param_groups"][0].items():
            self.param_groups[0][k] = v
