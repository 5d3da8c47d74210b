"""Checkpoint / wire compatibility -- SURVEY.md §8f rank 4.

Reference: ``utils/__init__.py:60-83`` (``extract_model_state_dict`` / ``load_ckpt``), ``utils/save_weights_only.py:14-17``
and the ``--pt_model`` prefix filter of ``train.py:22-33``.  A SinNeRF (pytorch-lightning) checkpoint is
``{'state_dict': {'nerf_coarse.xyz_encoding_1.0.weight': ..., 'nerf_fine....': ..., <discriminator / ViT keys>}}``;
``sinnerf_amd.NeRF`` has the reference's parameter names, so these helpers load reference checkpoints into it unchanged
(``eval.py:139-140``: ``load_ckpt(nerf_coarse, ckpt_path, model_name='nerf_coarse')``).
"""
import torch


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=()):
    """``utils/__init__.py:60-75``.  ``ckpt_path`` may also be an already loaded dict."""
    checkpoint = ckpt_path if isinstance(ckpt_path, dict) else torch.load(ckpt_path, map_location=torch.device("cpu"))
    checkpoint_ = {}
    if "state_dict" in checkpoint:                      # a pytorch-lightning checkpoint
        checkpoint = checkpoint["state_dict"]
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        for prefix in prefixes_to_ignore:
            if k.startswith(prefix):
                print("ignore", k)
                break
        else:
            checkpoint_[k] = v
    return checkpoint_


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=()):
    """``utils/__init__.py:78-83``."""
    model_dict = model.state_dict()
    model_dict.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(model_dict)


def load_nerf_only(system, ckpt_path):
    """``train.py:22-33`` with ``--nerf_only``: warm-start ``nerf_coarse`` / ``nerf_fine`` from a SinNeRF checkpoint."""
    load_ckpt(system.nerf_coarse, ckpt_path, model_name="nerf_coarse")
    if hasattr(system, "nerf_fine"):
        load_ckpt(system.nerf_fine, ckpt_path, model_name="nerf_fine")


def save_weights_only(system_or_state_dict, path):
    """``utils/save_weights_only.py:14-17``: store the bare ``state_dict`` (keys keep the ``nerf_coarse.`` / ``nerf_fine.``
    prefixes Lightning would write, so ``extract_model_state_dict`` and the reference's ``eval.py`` read it back)."""
    sd = system_or_state_dict if isinstance(system_or_state_dict, dict) else system_or_state_dict.state_dict()
    torch.save({k: v.detach().cpu() for k, v in sd.items()}, path)
