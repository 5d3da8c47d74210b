#!/usr/bin/env python3
"""Stage the UNMODIFIED reference hot-path modules where the GPU box can import them (TEST INFRASTRUCTURE, like the rest of
``oracle/``).

``/root/reference`` exists in the build container only.  ``__graft_entry__.build()`` runs this script there: it copies
``models/rendering.py``, ``models/nerf.py`` and ``models/activations.py`` (the whole hot path, SURVEY.md §8a: they import
nothing but torch / numpy) byte-for-byte into ``oracle/_ref/models/`` and records their sha256 in ``oracle/_ref/MANIFEST.json``.
``oracle/_ref/`` is git-ignored (reference sources never enter the history) but NOT gpurun-ignored, so it travels to the GPU
box with the snapshot exactly like the built ``.so``.  Consumers: ``bench.py``'s ``cpu_baseline`` leg (``kind: "reference"``
when the staged copy is present, ``"port"`` = ``oracle/torch_ref.py`` otherwise) and ``tests/test_ref_stage_cpu.py``
(staged reference == port ≤ 1 ulp).  Nothing under ``sinnerf_amd/`` may import it.

usage: python oracle/stage_ref.py [reference_root]        (default /root/reference; exit 0 and a note when it is absent)
"""
import hashlib
import importlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
FILES = ("models/rendering.py", "models/nerf.py", "models/activations.py")
# The reference's own CALLERS of the path (VERDICT r3 "prove the drop-in with the reference's own callers"): eval.py
# (batched_inference, eval.py:84-115) and models/sinnerf.py (SinNeRF.forward, sinnerf.py:171-193) with the import closure
# they drag in -- staged the same way (byte-for-byte, git-ignored), imported by tests/test_ref_callers*.py only, with stub
# modules standing in for the third-party packages this image lacks (install_stubs below).
CALLER_FILES = ("eval.py", "losses.py", "metrics.py", "models/sinnerf.py", "models/discriminator.py", "models/diff_aug.py",
                "models/extractor.py", "utils/__init__.py", "utils/optimizers.py", "utils/save_weights_only.py",
                "utils/visualization.py", "utils/warmup_scheduler.py", "datasets/__init__.py",
                "datasets/blender_ray_patch_1image_proj.py", "datasets/blender_ray_patch_1image_rot3d.py", "datasets/depth_utils.py",
                "datasets/dtu_proj.py", "datasets/llff.py", "datasets/llff_ray_patch_1image_proj.py", "datasets/ray_utils.py")
STUB_ROOTS = ("cv2", "torchvision", "kornia", "pytorch_lightning", "imageio", "piq", "test_tube", "timm", "lpips")


def stage(ref_root="/root/reference"):
    """Copy FILES from ``ref_root`` into oracle/_ref/; returns the manifest dict, or None when the reference is absent."""
    if not all(os.path.isfile(os.path.join(ref_root, f)) for f in FILES):
        return None
    os.makedirs(os.path.join(DEST, "models"), exist_ok=True)
    man = {"source": ref_root, "files": {}}
    for f in FILES + tuple(c for c in CALLER_FILES if os.path.isfile(os.path.join(ref_root, c))):
        dst = os.path.join(DEST, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(ref_root, f), dst)
        man["files"][f] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    # `models` must be a package for nerf.py's `from models.activations import ...`; the reference's own __init__ is empty
    open(os.path.join(DEST, "models", "__init__.py"), "w").close()
    json.dump(man, open(os.path.join(DEST, "MANIFEST.json"), "w"), indent=1)
    return man


def available():
    return all(os.path.isfile(os.path.join(DEST, f)) for f in FILES)


def load():
    """Import the staged reference modules; returns (rendering, nerf) module objects.  ``models`` is resolved from
    oracle/_ref FIRST and removed from sys.path afterwards (the modules stay cached under their reference names)."""
    if not available():
        raise FileNotFoundError("oracle/_ref is not staged (run oracle/stage_ref.py where /root/reference exists)")
    for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
        mod = sys.modules[name]
        if not (getattr(mod, "__file__", "") or "").startswith(DEST):
            del sys.modules[name]
    sys.path.insert(0, DEST)
    try:
        rendering = importlib.import_module("models.rendering")
        nerf = importlib.import_module("models.nerf")
    finally:
        sys.path.remove(DEST)
    return rendering, nerf


def callers_available():
    return available() and all(os.path.isfile(os.path.join(DEST, f)) for f in CALLER_FILES)


class _StubThing:
    """what a stub module hands out for any attribute: constructible, callable, subscriptable, attribute-transparent"""
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return self
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _StubThing()
    def __getitem__(self, k): return _StubThing()
    def __iter__(self): return iter(())


class _StubFinder:
    """meta-path finder fabricating the third-party packages of STUB_ROOTS that this image lacks: `import cv2`,
    `from torchvision import transforms as T`, `from kornia.losses import ssim` ... all succeed and yield inert objects.
    pytorch_lightning.LightningModule is a real torch.nn.Module subclass (SinNeRF derives from it)."""
    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        import types
        import torch

        class _Mod(types.ModuleType):
            def __getattr__(self, attr):
                if attr.startswith("__"):
                    raise AttributeError(attr)
                v = type(attr, (_StubThing,), {})
                setattr(self, attr, v)
                return v
        m = _Mod(spec.name)
        m.__path__ = []
        m.__stub__ = True
        if spec.name == "pytorch_lightning":
            m.LightningModule = type("LightningModule", (torch.nn.Module,), {})
        return m

    def exec_module(self, module):
        pass


def install_stubs():
    """stub every STUB_ROOTS package that cannot be imported here; returns the list of stubbed roots"""
    import importlib.util
    missing = []
    for r in STUB_ROOTS:
        if r in sys.modules and not getattr(sys.modules[r], "__stub__", False):
            continue
        try:
            found = importlib.util.find_spec(r) is not None and not getattr(sys.modules.get(r), "__stub__", False)
        except (ImportError, ValueError):
            found = False
        if not found:
            missing.append(r)
    if missing and not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder(missing))
    return missing


_REF_TOP = ("models", "eval", "losses", "metrics", "utils", "datasets")


def load_callers(patch=None):
    """Import the UNMODIFIED reference callers from oracle/_ref: returns (eval module, models.sinnerf module).

    ``patch``: None = the reference's own path underneath (CPU check of this harness), or an object with ``render_rays``,
    ``NeRF``, ``Embedding`` (e.g. the ``sinnerf_amd`` package) installed exactly as INTEGRATION.md §1 prescribes -- assigned
    onto ``models.rendering`` / ``models.nerf`` BEFORE eval.py / models/sinnerf.py are imported, since both bind the names at
    import time (eval.py:9-10, sinnerf.py:11-13).  The staged tree shadows site-packages for the reference's top-level names
    (HuggingFace ``datasets`` is installed here) only while this function imports."""
    if not callers_available():
        raise FileNotFoundError("oracle/_ref does not hold the reference callers (run oracle/stage_ref.py where /root/reference exists)")
    install_stubs()
    saved = {}
    for name in list(sys.modules):
        if name.split(".")[0] in _REF_TOP:
            saved[name] = sys.modules.pop(name)
    sys.path.insert(0, DEST)
    try:
        rendering = importlib.import_module("models.rendering")
        nerf = importlib.import_module("models.nerf")
        if patch is not None:
            rendering.render_rays = patch.render_rays                     # INTEGRATION.md §1, second form
            nerf.NeRF, nerf.Embedding = patch.NeRF, patch.Embedding
        ev = importlib.import_module("eval")
        sn = importlib.import_module("models.sinnerf")
    finally:
        sys.path.remove(DEST)
        for name in list(sys.modules):                                    # leave no shadowing `datasets` / `utils` behind
            if name.split(".")[0] in _REF_TOP:
                del sys.modules[name]
        sys.modules.update(saved)
    return ev, sn


def discriminator_available():
    return available() and all(os.path.isfile(os.path.join(DEST, f)) for f in ("models/discriminator.py", "models/diff_aug.py"))


def load_discriminator():
    """The reference's UNMODIFIED ``models/discriminator.py`` (+ ``models/diff_aug.py``; both need torch / numpy only) from
    oracle/_ref: returns the module (``.Discriminator``, ``.DiffAugment``).  The discriminator is NOT on the hot path -- north_star
    keeps it on stock PyTorch -- it is the reference-side consumer of the rendered side patch (``models/sinnerf.py:143-145, 445-487``)
    that BASELINE configs[2] "full SinNeRF losses" names; bench.py's ``train_cfg3_full`` leg and tests/test_full_losses_gpu.py plug it
    into ``SinNeRFSystem.side_loss``."""
    if not discriminator_available():
        raise FileNotFoundError("oracle/_ref does not hold models/discriminator.py (run oracle/stage_ref.py where /root/reference exists)")
    load()                                              # makes `models` resolve to the staged package
    sys.path.insert(0, DEST)
    try:
        return importlib.import_module("models.discriminator")
    finally:
        sys.path.remove(DEST)


def build_reference_models(params_list):
    """The reference's own ``NeRF`` / ``Embedding`` objects carrying the given state dicts (list of {name: ndarray})."""
    import torch
    _, nerf = load()
    models = []
    for p in params_list:
        m = nerf.NeRF(use_new_activation=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        models.append(m.eval())
    return models, [nerf.Embedding(3, 10), nerf.Embedding(3, 4)]


if __name__ == "__main__":
    m = stage(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    print("staged %d reference files into %s" % (len(m["files"]), DEST) if m else "reference not present: nothing staged")
