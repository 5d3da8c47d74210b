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
        """``torch.optim.Adam``'s own layout ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}, per
        parameter in parameter order), so a checkpoint written here resumes under the reference's optimiser
        (``utils/__init__.py:19-21``) and the other way round."""
        g = self.param_groups[0]
        params = self.grads.params
        state, off = {}, 0
        if self.step_count > 0:
            for i, p in enumerate(params):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
                off += n
        group = {k: v for k, v in g.items() if k != "params"}
        group["params"] = list(range(len(params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state):
        """Accepts the ``torch.optim.Adam`` layout (the reference's / Lightning's ``optimizer_states`` entry, or
        ``state_dict()`` above) and the round-2 private layout (top-level ``step`` / ``exp_avg`` / ``exp_avg_sq``).  Shapes are
        checked per parameter: state of a different or re-ordered parameter set is refused."""
        params = self.grads.params
        if "state" in state:
            st = state["state"]
            groups = state["param_groups"]
            ids = [i for g in groups for i in g["params"]]
            if len(ids) != len(params):
                raise ValueError("FlatAdam.load_state_dict: optimizer state has %d parameters, this model has %d" % (len(ids), len(params)))
            if len(st) not in (0, len(params)):
                raise ValueError("FlatAdam.load_state_dict: partial per-parameter state (%d of %d)" % (len(st), len(params)))
            steps, off = set(), 0
            for i, p in zip(ids, params):
                n = p.numel()
                if st:
                    e = st[i] if i in st else st[str(i)]
                    if tuple(e["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError("FlatAdam.load_state_dict: state %s has shape %s, parameter has %s -- a different or "
                                         "re-ordered parameter set" % (i, tuple(e["exp_avg"].shape), tuple(p.shape)))
                    self.exp_avg[off:off + n].copy_(e["exp_avg"].to(self.flat.device).reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].to(self.flat.device).reshape(-1))
                    steps.add(int(e["step"]))
                off += n
            if len(steps) > 1:
                raise ValueError("FlatAdam.load_state_dict: per-parameter step counts differ (%s): one fused step has one count" % sorted(steps))
            if not st:
                self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            self.step_count = steps.pop() if steps else 0
            src = groups[0]
        else:
            if state["exp_avg"].numel() != self.flat.numel():
                raise ValueError("FlatAdam.load_state_dict: optimizer state belongs to a different parameter set")
            self.step_count = int(state["step"])
            self.exp_avg.copy_(state["exp_avg"].to(self.flat.device).reshape(-1))
            self.exp_avg_sq.copy_(state["exp_avg_sq"].to(self.flat.device).reshape(-1))
            src = state["param_groups"][0]
        for k, v in src.items():
            if k != "params":
                self.param_groups[0][k] = v
