"""System surface around the hot path -- the hooks ``train.py`` / Lightning 0.10 call on ``SinNeRF``
(reference ``models/sinnerf.py:124-586``), restricted to what touches the accelerated path.

In scope (SURVEY.md §8b "System surface"): ``nerf_coarse`` / ``nerf_fine`` / ``models`` / ``embeddings`` attributes
(``sinnerf.py:133-141``, used by ``train.py:29-30``), ``forward(rays)`` with the ray-chunk loop (``sinnerf.py:171-193``),
``configure_optimizers`` (``sinnerf.py:202-210`` with ``utils/__init__.py:11-57`` defaults: Adam eps=1e-8, MultiStepLR),
an MSE(+SmoothL1-depth) ``training_step`` (``losses.py:12-22``, ``sinnerf.py:310-319``) and ``validation_step`` PSNR
(``sinnerf.py:556-577``, ``metrics.py:5-15``).  Out of scope and kept on stock PyTorch by ``north_star``: datasets /
dataloaders, discriminator + DiffAugment, DINO-ViT loss, warping helpers, TensorBoard logging -- a full SinNeRF run plugs
those in around this class exactly as the reference does around its own.

pytorch_lightning 0.10.0 is not installable offline, so the class is a plain ``nn.Module`` exposing the same hook names
and return shapes (duck-type compatible with the PL 0.10 Trainer loop); ``train_step`` below is the minimal driver used by
tests and benchmarks (zero -> forward -> loss -> backward -> one flat all-reduce -> Adam).
"""
from collections import defaultdict
from types import SimpleNamespace

import torch
from torch import nn

from .nerf import Embedding, NeRF
from .optim import FlatAdam
from .parallel import all_reduce_mean_grads, broadcast_parameters
from .losses import psnr, render_loss      # noqa: F401  (psnr re-exported: metrics.py:14-15)
from .rendering import render_rays

DEFAULT_HPARAMS = dict(N_samples=64, N_importance=64, use_disp=False, perturb=1.0, noise_std=1.0, chunk=32 * 1024,
                       lr=5e-4, weight_decay=0.0, decay_step=[20], decay_gamma=0.1, depth_weight=0.0, white_back=True,
                       compute_dtype="fp32",
                       # the adversarial term of the reference's second training stage (opt.py:98,107; README "Step 2": 0.01).  It only
                       # takes effect once a discriminator module is attached (attach_discriminator): the discriminator itself stays on
                       # stock PyTorch (north_star) and is not part of this package
                       dis_weight=0.0, dloss="hinge", patch_hw=None)




def rebuild_scheduler(sch, new_opt):
    """The same learning-rate schedule, at the same epoch, on another optimiser.

    A ``MultiStepLR`` is constructed fresh (its constructor takes one initial step: ``last_epoch`` 0, the group's current lr kept)
    and then given the old one's ``state_dict`` -- epoch counter, step count, ``_last_lr``, milestones.  Passing
    ``last_epoch=sch.last_epoch`` to the constructor instead would resume one epoch AHEAD (the initial step increments it), so every
    milestone of ``utils/__init__.py:34-36`` would fire one step early.  Other scheduler types are re-pointed in place."""
    if isinstance(sch, torch.optim.lr_scheduler.MultiStepLR):
        lrs = [g["lr"] for g in new_opt.param_groups]
        ns = torch.optim.lr_scheduler.MultiStepLR(new_opt, milestones=sorted(sch.milestones.elements()), gamma=sch.gamma)
        ns.load_state_dict(sch.state_dict())
        for g, lr in zip(new_opt.param_groups, lrs):
            g["lr"] = lr
        assert ns.last_epoch == sch.last_epoch
        sch.optimizer = new_opt       # a caller still holding the old object at least drives the live optimiser's lr
        return ns
    sch.optimizer = new_opt
    return sch


class SinNeRFSystem(nn.Module):
    def __init__(self, hparams=None, **kw):
        super().__init__()
        hp = dict(DEFAULT_HPARAMS)
        hp.update(vars(hparams) if hasattr(hparams, "__dict__") else (hparams or {}))
        hp.update(kw)
        self.hparams = SimpleNamespace(**hp)
        self.embedding_xyz = Embedding(3, 10)                                    # sinnerf.py:133
        self.embedding_dir = Embedding(3, 4)                                     # sinnerf.py:134
        self.embeddings = [self.embedding_xyz, self.embedding_dir]
        self.nerf_coarse = NeRF(use_new_activation=True, compute_dtype=hp["compute_dtype"])   # sinnerf.py:137
        self.models = [self.nerf_coarse]
        if hp["N_importance"] > 0:
            self.nerf_fine = NeRF(use_new_activation=True, compute_dtype=hp["compute_dtype"])  # sinnerf.py:140
            self.models.append(self.nerf_fine)
        self.white_back = hp["white_back"]       # dataset property in the reference (blender/dtu True, llff False)
        self._flat = None
        self.D = None                            # sinnerf.py:143-145: built by the reference when dis_weight > 0; here: attach_discriminator()

    # ---- sinnerf.py:143-145, 207-208, 445-471 (dloss == 'hinge'): the hooks the discriminator plugs into --------------------------------
    def attach_discriminator(self, D, dis_weight=None, patch_hw=None):
        """Plug a discriminator (any ``nn.Module`` mapping a ``(1, 3, psx, psy)`` patch to logits -- the reference's
        ``models/discriminator.py::Discriminator`` unchanged) into the patch step, as ``sinnerf.py:143-145`` does when
        ``dis_weight > 0``.  ``patch_hw`` = (psx, psy) of the side patch (``real_patch.shape[-2:]``, sinnerf.py:281) when the batch
        carries no ``real_patch``.  Call before ``configure_optimizers``: it then also returns ``opt_d`` (sinnerf.py:207-208)."""
        self.D = D
        if dis_weight is not None:
            self.hparams.dis_weight = dis_weight
        if patch_hw is not None:
            self.hparams.patch_hw = tuple(patch_hw)
        if self.hparams.dloss != "hinge":
            raise NotImplementedError("dloss=%r: only the reference's default 'hinge' (opt.py:98) is mirrored" % (self.hparams.dloss,))
        return self

    def _side_patch(self, results_side, batch):
        """``rearrange(results_side['rgb_fine'], '(b p q) c -> b c p q')`` of sinnerf.py:327-330 for b = 1"""
        hw = tuple(batch["real_patch"].shape[-2:]) if "real_patch" in batch else self.hparams.patch_hw
        if hw is None:
            raise RuntimeError("the discriminator needs the patch shape: pass patch_hw=(psx, psy) or a batch with 'real_patch'")
        rgb = results_side["rgb_fine"]
        return rgb.reshape(1, hw[0], hw[1], 3).permute(0, 3, 1, 2)

    def _generator_adv_loss(self, results_side, batch):
        """optimizer_idx == 0, hinge: ``loss_d = -mean(D(results_side['rgb_fine']))`` (sinnerf.py:446-450), weighted by
        ``dis_weight`` where the total is formed (sinnerf.py:499)"""
        return -torch.mean(self.D(self._side_patch(results_side, batch))) * self.hparams.dis_weight

    def discriminator_step(self, batch):
        """optimizer_idx == 1 of ``training_step`` (sinnerf.py:462-471, hinge): the discriminator sees the real patch and the
        DETACHED side render, so the only work on the accelerated path is ONE no-grad render of ``rays_side`` (PL 0.10 calls
        ``training_step`` once per optimiser: the reference re-runs all four renders here and back-propagates through them for
        gradients the discriminator's optimiser never uses).  Returns ``{'loss': loss_d}`` for ``opt_d``."""
        if self.D is None:
            raise RuntimeError("no discriminator attached (attach_discriminator)")
        with torch.no_grad():
            results_side = self(batch["rays_side"].reshape(-1, 8))
        fake = self._side_patch(results_side, batch).detach()
        real = batch["real_patch"]
        pred_real, pred_fake = self.D(real), self.D(fake)
        loss_real = torch.relu(torch.ones_like(pred_real) - pred_real).mean()
        loss_gen = torch.relu(torch.ones_like(pred_fake) + pred_fake).mean()
        loss_d = (loss_real + loss_gen) / 2
        # sinnerf.py:499: the total the reference back-propagates in the optimizer_idx == 1 pass carries loss_d * dis_weight
        # (Adam is almost, not exactly, scale-invariant -- eps, weight_decay coupling); the raw value goes to the log
        return {"loss": loss_d * self.hparams.dis_weight, "log": {"train/loss_d": loss_d.detach()}}

    # ---- sinnerf.py:171-193 -------------------------------------------------------------------------------------
    def forward(self, rays):
        B = rays.shape[0]
        results = defaultdict(list)
        hp = self.hparams
        for i in range(0, B, hp.chunk):
            chunk_res = render_rays(self.models, self.embeddings, rays[i:i + hp.chunk], hp.N_samples, hp.use_disp,
                                    hp.perturb, hp.noise_std, hp.N_importance, hp.chunk, self.white_back)
            for k, v in chunk_res.items():
                results[k].append(v)
        # one chunk (every training batch): hand the tensors on -- torch.cat of a single tensor is a device copy per key
        return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in results.items()}

    # ---- sinnerf.py:202-210 + utils/__init__.py:11-57 -----------------------------------------------------------
    def configure_optimizers(self):
        """Adam(lr, eps=1e-8, weight_decay) + MultiStepLR as ``get_optimizer`` / ``get_scheduler`` build them
        (utils/__init__.py:19-21, 27-31).  The optimiser is ``FlatAdam`` (a ``torch.optim.Optimizer``): parameters and
        gradients of both NeRFs live in flat buffers, one step = the single all-reduce + one ``sn_adam_step`` launch."""
        hp = self.hparams
        params = [p for m in self.models for p in m.parameters()]
        if params[0].is_cuda:
            self.optimizer = FlatAdam(self.models, lr=hp.lr, eps=1e-8, weight_decay=hp.weight_decay)
            self._flat = self.optimizer.grads
        else:       # module still on the host (Lightning calls configure_optimizers before .to(device) in some versions):
            # the reference's own optimiser; the render path itself has no CPU form and raises on CPU tensors
            self.optimizer = torch.optim.Adam(params, lr=hp.lr, eps=1e-8, weight_decay=hp.weight_decay)
        scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=hp.decay_step, gamma=hp.decay_gamma)
        self._schedulers = [scheduler]
        self.__dict__.pop("_step_graphs", None)      # captured steps bake in the old flat-buffer addresses
        opts = [self.optimizer]
        if self.D is not None and hp.dis_weight > 0:                             # sinnerf.py:207-208: get_optimizer(hparams, [self.D], rate=0.2)
            self.opt_d = torch.optim.Adam(self.D.parameters(), lr=hp.lr * 0.2, eps=1e-8, weight_decay=hp.weight_decay)
            opts.append(self.opt_d)
        return opts, [scheduler]

    # ---- losses.py:12-22 (MSE coarse + fine) + SL1Loss of depth_fine and depth_coarse (sinnerf.py:32-42, 310-319):
    #      value, gradients and PSNR from the fused sn_render_loss kernel pair (sinnerf_amd/losses.py)
    def loss(self, results, rgbs, depths=None, with_stats=False):
        use_depth = depths is not None and self.hparams.depth_weight > 0
        total, stats = render_loss(results, rgbs, depths if use_depth else None, w_depth=self.hparams.depth_weight)
        return (total, stats) if with_stats else total

    def training_step(self, batch, batch_idx=0, optimizer_idx=0):
        """One generator step on the accelerated path.

        * ``{"rays", "rgbs"[, "depths"]}``: one render, MSE coarse+fine (+ ``depth_weight`` x SmoothL1 depth) -- the core.
        * a SinNeRF patch batch (``rays_full`` present, the keys of ``models/sinnerf.py:277-299``): the FOUR renders of
          ``sinnerf.py:304-307`` -- ``rays`` (random rays of the reference view), ``rays_full`` (the strided patch with
          ground truth), ``rays_side`` (the patch seen from the unseen view), ``rays_proj`` (rays with projected depth) --
          with MSE on ``results`` and on the full patch (``patch_loss``, ``sinnerf.py:347``), SmoothL1 depth on
          ``results`` and ``results_proj`` (``:309-319``, ``useMask=False``), and ``self.side_loss(results_side, batch)`` for
          the unseen view.  In the reference that last term is the DINO-ViT feature loss / the discriminator, which
          ``north_star`` keeps on stock PyTorch: plug them in as ``side_loss``; the default (MSE against ``side_rgb``, the
          warped patch of ``sinnerf.py:300-302``) keeps the render and its backward on the step so that the timed work
          is the reference's.
        """
        if optimizer_idx == 1:
            return self.discriminator_step(batch)
        if "rays_full" in batch:
            return self._training_step_patches(batch)
        rays, rgbs = batch["rays"], batch["rgbs"]
        rays, rgbs = rays.reshape(-1, 8), rgbs.reshape(-1, 3)
        results = self(rays)
        loss, stats = self.loss(results, rgbs, batch.get("depths"), with_stats=True)
        p = stats["psnr_fine"] if "rgb_fine" in results else stats["psnr_coarse"]
        return {"loss": loss, "progress_bar": {"train_psnr": p}, "log": {"train/loss": loss.detach(), "train/psnr": p}}

    def side_loss(self, results_side, batch):
        """The unseen-view terms, which stay on PyTorch.  With a discriminator attached and ``dis_weight > 0``: the generator's
        adversarial loss of ``sinnerf.py:446-450``.  Otherwise a stand-in (MSE of the side render against the warped patch) that keeps
        the render and its backward on the step.  The DINO-ViT feature loss (``sinnerf.py:332-339``) needs weights from the network
        and is not runnable offline: assign your own callable (``system.side_loss = fn``) to add it."""
        if self.D is not None and self.hparams.dis_weight > 0:
            return self._generator_adv_loss(results_side, batch)
        tgt = batch.get("side_rgb")
        if tgt is None:
            return None
        total, _ = render_loss(results_side, tgt.reshape(-1, 3))
        return total

    def _training_step_patches(self, batch):
        hp = self.hparams
        r8 = lambda k: batch[k].reshape(-1, 8)
        results = self(r8("rays"))                                              # sinnerf.py:304
        results_full = self(r8("rays_full"))                                    # :305
        results_side = self(r8("rays_side"))                                    # :306
        results_proj = self(r8("rays_proj"))                                    # :307
        wd = hp.depth_weight
        depth = batch.get("depth")
        loss, stats = render_loss(results, batch["rgbs"].reshape(-1, 3), depth.reshape(-1) if (depth is not None and wd > 0) else None,
                                  w_depth=wd)                                    # :316 loss_g + :317-318 depth terms
        log = {"train/loss_g": loss.detach()}
        if "rgbs_full" in batch:                                                # :347 patch_loss(results_full, rgbs_full)
            l_full, _ = render_loss(results_full, batch["rgbs_full"].reshape(-1, 3))
            loss = loss + l_full
        if wd > 0 and "depth_proj" in batch:                                    # :309-311 SL1 on the projected rays
            l_proj, _ = render_loss(results_proj, None, batch["depth_proj"].reshape(-1), w_depth=wd)
            loss = loss + l_proj
            log["train/loss_depth_proj"] = l_proj.detach()
        l_side = self.side_loss(results_side, batch)
        if l_side is not None:
            loss = loss + l_side
        p = stats["psnr_fine"] if "rgb_fine" in results else stats["psnr_coarse"]
        log.update({"train/loss": loss.detach(), "train/psnr": p})
        return {"loss": loss, "progress_bar": {"train_psnr": p}, "log": log}

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        rays, rgbs = batch["rays"].reshape(-1, 8), batch["rgbs"].reshape(-1, 3)
        results = self(rays)
        loss, stats = self.loss(results, rgbs, with_stats=True)
        return {"val_loss": loss, "val_psnr": stats["psnr_fine"] if "rgb_fine" in results else stats["psnr_coarse"]}

    def validation_epoch_end(self, outputs):
        mean_psnr = torch.stack([x["val_psnr"] for x in outputs]).mean()
        return {"progress_bar": {"val_psnr": mean_psnr}, "log": {"val/psnr": mean_psnr}}

    # ---- minimal driver: one optimisation step with the single flat all-reduce -----------------------
    def _ensure_flat_optimizer(self):
        """``configure_optimizers`` may have run while the module was still on the host (stock Adam, no flat buffers, NO
        gradient exchange).  Once the parameters are on the device the optimiser is rebuilt as ``FlatAdam`` -- carrying over
        the hyper-parameters a scheduler may already have changed -- so that ``train_step`` always ends in the one flat
        all-reduce.  Adam state accumulated by a host-side optimiser is refused rather than silently dropped."""
        opt = getattr(self, "optimizer", None)
        if isinstance(opt, FlatAdam):
            return opt
        params = [p for m in self.models for p in m.parameters()]
        if not params[0].is_cuda:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                raise RuntimeError("SinNeRFSystem: a multi-rank step needs the parameters on a ROCm device (FlatAdam owns the "
                                   "gradient all-reduce); move the module with .to(device) first")
            if opt is None:
                self.configure_optimizers()
            return self.optimizer
        if opt is None:
            self.configure_optimizers()
            return self.optimizer
        if any(len(st) for st in opt.state.values()):
            raise RuntimeError("SinNeRFSystem: the host-side optimiser already holds Adam state; call configure_optimizers() "
                               "again after .to(device) (or load its state into the new FlatAdam) instead of stepping on")
        g = opt.param_groups[0]
        new = FlatAdam(self.models, lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"])
        if "initial_lr" in g:
            new.param_groups[0]["initial_lr"] = g["initial_lr"]
        # the scheduler is REBUILT on the new optimiser (patching `.optimizer` would leave its step-order bookkeeping --
        # the wrapper around optimizer.step, `_opt_called` -- bound to the discarded torch.optim.Adam): same milestones / gamma,
        # resumed at the epoch the old one had reached.  Lists configure_optimizers() returned earlier still hold the old pair:
        # re-read `self.optimizer` / `self._schedulers` (or call configure_optimizers() again) after .to(device)
        rebuilt = [rebuild_scheduler(sch, new) for sch in getattr(self, "_schedulers", [])]
        if rebuilt:
            self._schedulers = rebuilt
        self.optimizer, self._flat = new, new.grads
        self.__dict__.pop("_step_graphs", None)
        return new

    def setup_distributed(self):
        """Replicas start identical (what DDP's constructor does, train.py:51-52); returns the flat gradient buffer whose
        all-reduce is the step's one exchange."""
        broadcast_parameters(self.models)
        if self.D is not None:                   # the reference wraps the WHOLE LightningModule in DDP (train.py:51-52): D starts identical too
            broadcast_parameters([self.D])
        self._ensure_flat_optimizer()            # (an existing FlatAdam already saw the broadcast: its flat buffer IS p.data)
        if self._flat is None:
            raise RuntimeError("setup_distributed: parameters are on the host; move the module to a ROCm device first")
        return self._flat

    def _zero_forward_backward(self, batch):
        self.optimizer.zero_grad()
        out = self.training_step(batch)
        loss = out["loss"]
        # the root gradient, kept: backward() would fill a fresh ones_like every step.  READ-ONLY contract: autograd hands this
        # very tensor to the first backward nodes and to any hook / side_loss Function; an in-place op on it there (g.mul_(),
        # unscale / clip code) would corrupt every later step, so its version counter is checked and the tensor replaced if it
        # moved (eager steps; a captured graph replays the kernels recorded while it was still 1)
        unit, ver = self.__dict__.get("_unit_grad", (None, None))
        if (unit is None or unit.device != loss.device or unit.dtype != loss.dtype or unit.shape != loss.shape
                or unit._version != ver):
            unit = torch.ones_like(loss)
            self.__dict__["_unit_grad"] = (unit, unit._version)
        loss.backward(unit)
        if unit._version != self.__dict__["_unit_grad"][1]:
            raise RuntimeError("SinNeRFSystem: a backward hook / side_loss modified the root gradient in place; it is shared "
                               "between steps -- use out-of-place ops on incoming gradients")
        return out

    def train_step(self, batch, graph=False):
        """zero -> forward -> loss -> backward -> [all-reduce of the flat gradient buffer + fused Adam] (FlatAdam.step).

        ``graph=True``: the zero / forward / loss / backward part (~45 kernel launches and ~60 small torch ops per step, all
        on the current stream, nothing built or copied from the host) is captured ONCE into a HIP graph for this batch shape
        and replayed with the batch copied into static buffers; the exchange step stays eager (one all-reduce + one Adam
        launch), so the same code serves one rank and N ranks.  The returned dict holds the graph's static output tensors
        (overwritten by the next replay)."""
        self._ensure_flat_optimizer()
        if graph:
            out = self._graphed_step(batch)
        else:
            out = self._zero_forward_backward(batch)
        self.optimizer.step()                    # the one exchange step (RCCL all-reduce, mean) + sn_adam_step
        return out

    def train_step_adversarial(self, batch):
        """Both optimiser passes of one batch, as pytorch-lightning 0.10 drives ``training_step`` with two optimisers
        (``sinnerf.py:202-210, 271``): pass 0 -- the four renders with gradients, the generator's hinge term through the attached
        discriminator (its parameters frozen, as PL toggles ``requires_grad`` per optimiser), flat all-reduce + Adam on the NeRFs;
        pass 1 -- ``discriminator_step`` (one no-grad side render), backward through the discriminator only, ``opt_d.step()``.
        Returns (generator dict, discriminator dict)."""
        if self.D is None or not hasattr(self, "opt_d"):
            raise RuntimeError("train_step_adversarial: attach_discriminator() and configure_optimizers() first")
        d_params = list(self.D.parameters())
        for p in d_params:
            p.requires_grad_(False)
        try:
            out_g = self.train_step(batch)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        self.opt_d.zero_grad(set_to_none=True)
        out_d = self.discriminator_step(batch)
        out_d["loss"].backward()
        all_reduce_mean_grads(d_params)          # DDP averages D's gradients like every other parameter's (no-op at world size 1)
        self.opt_d.step()
        return out_g, out_d

    def _graphed_step(self, batch):
        tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
        hp = self.hparams
        # a captured step bakes in the flat parameter / gradient buffers and every hyper-parameter the launches read
        key = (tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in tensors.items())),
               self.optimizer.flat.data_ptr() if isinstance(self.optimizer, FlatAdam) else None,
               self._flat.flat.data_ptr() if self._flat is not None else None,
               hp.N_samples, hp.N_importance, hp.use_disp, hp.perturb, hp.noise_std, hp.chunk, hp.depth_weight,
               hp.compute_dtype, self.white_back, self.training)
        cache = self.__dict__.setdefault("_step_graphs", {})
        hit = cache.get(key)
        if hit is None:
            if not isinstance(self.optimizer, FlatAdam):
                raise RuntimeError("train_step(graph=True) needs the flat optimiser (parameters on a ROCm device)")
            static = {k: v.clone() for k, v in tensors.items()}
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):        # eager warm-up on a side stream: one-time attribute calls, pack tables,
                for _ in range(2):               # allocator pools -- nothing of that may happen during capture
                    self._zero_forward_backward(static)
            cur.wait_stream(side)
            for m in self.models:                # the re-pack of the weight blobs (one gather launch per blob, normally skipped
                m.invalidate_packed()            # while the parameters are unchanged) must be PART of the captured step:
            g = torch.cuda.CUDAGraph()           # the replays run behind optimizer steps no Python code sees
            # thread_local: an RCCL process group's watchdog thread polls events while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._zero_forward_backward(static)
            hit = cache[key] = (g, static, out)
        g, static, out = hit
        for k, v in tensors.items():
            static[k].copy_(v)
        g.replay()
        return out

    def replica_checksum(self):
        """fp64 sum and sum of squares of the flat parameter buffer: equal on every rank iff the replicas are identical
        (what the bench's multi-GPU training leg asserts after a few steps)."""
        flat = self.optimizer.flat.double() if isinstance(getattr(self, "optimizer", None), FlatAdam) else \
            torch.cat([p.detach().reshape(-1) for m in self.models for p in m.parameters()]).double()
        return torch.stack([flat.sum(), (flat * flat).sum()])
