"""Multi-GPU support of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Reference: the only parallelism in SinNeRF is Lightning DDP (``train.py:51-52``: ``distributed_backend='ddp'``): model
replicas, each rank renders its own patches, gradients are mean-reduced by DDP's bucketed all-reduce.  SURVEY.md §8e:

* inference: every ray is independent -> ``shard_rays`` partitions the (H*W, 8) ray array contiguously across ranks,
  **no collective** on the data path (an optional gather of the (N/R, 3) rgb tiles to rank 0 for display);
* training: replicas + **one** all-reduce per step of a single flat fp32 buffer holding the gradients of both NeRFs
  (1 191 688 floats = 4.77 MB).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 4.77 MB ring all-reduce is
  ~55 us, far below a >= 10 ms step, so it is issued once after backward, un-bucketed, on the compute stream.

``FlatGradBuffer`` makes every ``param.grad`` a view into the flat buffer, so autograd accumulates straight into it
and the all-reduce needs no gather/scatter copies.  Works with any backend (gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n items owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays, rank=None, world=None):
    """This rank's contiguous share of a ray tensor (N, 8) -- the inference partition (no collective needed)."""
    if rank is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_bounds(rays.shape[0], rank, world)
    return rays[lo:hi]


def gather_rows(local, n_total, dst=0):
    """Optional: collect per-rank row blocks (e.g. (N/R, 3) rgb tiles) on ``dst`` in ray order.  Returns the full
    tensor on ``dst`` and None elsewhere.  Not on the timed path."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], 0)


class FlatGradBuffer:
    """One flat fp32 gradient buffer for a list of modules; ``param.grad`` are views into it."""

    def __init__(self, modules, sink=True):
        modules = list(modules)
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        # Gradient sink: the weight-gradient finish kernel of a NeRF whose parameters all live here accumulates straight into
        # these views (sn_weight_grads(accumulate=1), sinnerf_amd/autograd.py) instead of returning 24 tensors per network
        # for autograd's AccumulateGrad to add one launch at a time.  Per-parameter autograd hooks do not fire in that mode
        # (this class replaces DDP's hook-driven bucketing by one explicit all-reduce, so none are needed); it is taken only
        # while every .grad still IS its view (sinnerf_amd.autograd._sink_of).  Functional differentiation w.r.t. the parameters
        # (torch.autograd.grad(loss, params)) needs the gradients RETURNED: build the buffer with sink=False for that.
        if sink:
            for m in modules:
                if hasattr(m, "raw_tensors") and all(t.requires_grad for t in m.raw_tensors()):
                    m._grad_sink = [t.grad for t in m.raw_tensors()]

    def zero(self):
        """Replaces ``optimizer.zero_grad()``: one memset, views stay attached."""
        self.flat.zero_()
        self._reattach()

    def _slot(self, i):
        off = self._offsets[i]
        p = self.params[i]
        return self.flat[off:off + p.numel()].view_as(p)

    @property
    def _offsets(self):
        if not hasattr(self, "_off"):
            self._off, o = [], 0
            for p in self.params:
                self._off.append(o)
                o += p.numel()
        return self._off

    def _is_view(self, g, i):
        p = self.params[i]
        return (g is not None and g.data_ptr() == self.flat.data_ptr() + self._offsets[i] * self.flat.element_size()
                and g.shape == p.shape and g.is_contiguous() and g.dtype == self.flat.dtype)

    def _reattach(self):
        for i, p in enumerate(self.params):
            if not self._is_view(p.grad, i):
                p.grad = self._slot(i)

    def sync_views(self):
        """Make the flat buffer hold the gradients autograd actually produced.  ``param.grad`` normally IS a view into it
        (autograd accumulates in place), but ``module.zero_grad()`` / ``optimizer.zero_grad()`` with PyTorch's default
        ``set_to_none=True`` detach the views: autograd then allocates fresh ``.grad`` tensors and the flat buffer would go
        stale (all-reduce and optimizer step on zeros -- training silently stops).  Stray gradients are copied into their
        slot, a ``None`` gradient (parameter unused this step) becomes zeros, and the views are re-attached.  Called by
        ``all_reduce_mean`` and ``FlatAdam.step``.  Returns the number of parameters that had to be repaired."""
        fixed = 0
        for i, p in enumerate(self.params):
            g = p.grad
            if self._is_view(g, i):
                continue
            slot = self._slot(i)
            if g is None:
                slot.zero_()
            else:
                slot.copy_(g.reshape(p.shape))
            p.grad = slot
            fixed += 1
        return fixed

    def all_reduce_mean(self, group=None):
        """The single exchange step of a training iteration: sum over ranks, then scale by 1/world (DDP semantics)."""
        self.sync_views()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            prof = getattr(self, "profile", None)        # bench.py: a list collects (start, end) events around the collective
            if prof is not None and self.flat.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / dist.get_world_size(group))
            if prof is not None and self.flat.is_cuda:
                e1.record()
                prof.append((e0, e1))
        return self.flat


def broadcast_parameters(modules, src=0):
    """Make replicas identical at start-up (what DDP's constructor does)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=src)
            if hasattr(m, "invalidate_packed"):          # written through .data: Parameter._version did not move
                m.invalidate_packed()


def all_reduce_mean_grads(params):
    """Average the ``.grad`` of stock-PyTorch side modules over the ranks before their optimiser steps -- what DDP does for every
    parameter of the LightningModule, the discriminator included (``train.py:51-52``, ``sinnerf.py:202-210``).  ONE flat all-reduce
    of the coalesced gradients; a no-op at world size 1.  Parameters without a gradient contribute zeros (DDP's unused-parameter
    behaviour) so that every rank reduces the same buffer."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n

