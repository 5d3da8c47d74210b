"""oracle/_ref = the unmodified reference hot-path modules staged by oracle/stage_ref.py: bench.py's
cpu_baseline times THEM (kind "reference").  Here: the staged copy is byte-identical to /root/reference when that exists,
its render_rays agrees with the torch port (oracle/torch_ref.py) to float rounding, and the product package never imports it."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_np as O          # noqa: E402
from oracle import stage_ref, torch_ref    # noqa: E402


def _staged():
    if not stage_ref.available():
        if stage_ref.stage() is None:
            pytest.skip("no /root/reference here and oracle/_ref was not staged by build()")
    return stage_ref.load()


def test_staged_files_are_byte_identical_to_the_reference():
    _staged()
    man = json.load(open(os.path.join(stage_ref.DEST, "MANIFEST.json")))
    for f, sha in man["files"].items():
        assert hashlib.sha256(open(os.path.join(stage_ref.DEST, f), "rb").read()).hexdigest() == sha
        src = os.path.join("/root/reference", f)
        if os.path.isfile(src):
            assert open(src, "rb").read() == open(os.path.join(stage_ref.DEST, f), "rb").read()


def test_staged_reference_render_equals_port():
    rendering, _ = _staged()
    params = [O.init_params(s, teacher=True) for s in (0, 1)]
    models, emb = stage_ref.build_reference_models(params)
    rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::700][:192].copy())
    torch.set_num_threads(4)
    with torch.no_grad():
        ref = rendering.render_rays(models, emb, rays, 64, False, 0, 0, 64, 32768, True)
        port = torch_ref.render([{k: torch.from_numpy(v) for k, v in p.items()} for p in params], rays, 64, 64, True)
    for k in ("rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine"):
        a, b = ref[k].numpy(), port[k].numpy()
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(a).max()), (k, np.abs(a - b).max())


def test_product_package_never_imports_the_oracle():
    out = subprocess.run(["grep", "-rlE", r"oracle|_ref\b|stage_ref", os.path.join(REPO, "sinnerf_amd"), "--include=*.py",
                          "--include=*.hip", "--include=*.h"], capture_output=True, text=True)
    assert out.stdout.strip() == "", out.stdout
