"""Checkpoint / wire compatibility -- SURVEY.md §8f rank 4.

What the reference's helpers do (``utils/__init__.py:60-83``, ``utils/save_weights_only.py:14-17``, the ``--pt_model`` prefix
filter of ``train.py:22-33``) and what this module therefore has to understand: a SinNeRF (pytorch-lightning) checkpoint is
``{'state_dict': {'nerf_coarse.xyz_encoding_1.0.weight': ..., 'nerf_fine....': ..., <discriminator / ViT keys>}}``, or the bare
inner dict when written by ``save_weights_only``.  ``sinnerf_amd.NeRF`` keeps the reference's parameter names, so selecting the
entries of one sub-module and stripping its ``"<name>."`` prefix is all that is needed to load a reference checkpoint
(``eval.py:139-140``: ``load_ckpt(nerf_coarse, ckpt_path, model_name='nerf_coarse')``).
"""
import torch


def _flat_state(ckpt):
    """path | loaded checkpoint | bare state dict  ->  the flat ``{qualified name: tensor}`` mapping"""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    return ckpt.get("state_dict", ckpt)


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=()):
    """The sub-module ``model_name``'s entries with the ``"<model_name>."`` prefix removed; entries whose remaining name starts
    with one of ``prefixes_to_ignore`` are dropped and reported, as the reference's helper of the same name does
    (``utils/__init__.py:60-75``; same selection rule: ``str.startswith(model_name)``, then ``len(model_name) + 1`` characters cut)."""
    skip = tuple(prefixes_to_ignore)
    cut = len(model_name) + 1
    picked = {}
    for name, tensor in _flat_state(ckpt_path).items():
        if not name.startswith(model_name):
            continue
        local = name[cut:]
        if skip and local.startswith(skip):
            print("ignore", local)
            continue
        picked[local] = tensor
    return picked


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=()):
    """Overlay the checkpoint's entries for ``model_name`` on the model's current state (parameters the checkpoint lacks keep
    their values, unknown entries fail in ``load_state_dict`` -- ``utils/__init__.py:78-83``)."""
    merged = dict(model.state_dict())
    merged.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(merged)


def load_nerf_only(system, ckpt_path):
    """``train.py:22-33`` with ``--nerf_only``: warm-start ``nerf_coarse`` / ``nerf_fine`` from a SinNeRF checkpoint."""
    state = _flat_state(ckpt_path)                     # read the file once
    load_ckpt(system.nerf_coarse, state, model_name="nerf_coarse")
    if hasattr(system, "nerf_fine"):
        load_ckpt(system.nerf_fine, state, model_name="nerf_fine")


def save_weights_only(system_or_state_dict, path):
    """``utils/save_weights_only.py:14-17``: store the bare ``state_dict`` (keys keep the ``nerf_coarse.`` / ``nerf_fine.``
    prefixes Lightning would write, so ``extract_model_state_dict`` and the reference's ``eval.py`` read it back)."""
    sd = system_or_state_dict if isinstance(system_or_state_dict, dict) else system_or_state_dict.state_dict()
    torch.save({k: v.detach().cpu() for k, v in sd.items()}, path)
