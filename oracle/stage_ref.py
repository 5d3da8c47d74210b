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


def stage(ref_root="/root/reference"):
    """Copy FILES from ``ref_root`` into oracle/_ref/; returns the manifest dict, or None when the reference is absent."""
    if not all(os.path.isfile(os.path.join(ref_root, f)) for f in FILES):
        return None
    os.makedirs(os.path.join(DEST, "models"), exist_ok=True)
    man = {"source": ref_root, "files": {}}
    for f in FILES:
        dst = os.path.join(DEST, f)
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
