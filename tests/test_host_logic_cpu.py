"""Host-side logic that needs no GPU: configuration refusals (there is no torch-op second backend), the learning-rate schedule
surviving the optimiser upgrade of ``SinNeRFSystem._ensure_flat_optimizer``, the checkpoint key filter."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def test_other_layer_configurations_are_refused_at_construction():
    """models/nerf.py:47-50 accepts any D / W / skips / input widths; only the one both call sites build exists in HIP."""
    import sinnerf_amd
    sinnerf_amd.NeRF()                                                       # the defaults ARE the supported configuration
    sinnerf_amd.NeRF(8, 256, 63, 27, [4], True)
    for kw in (dict(D=4), dict(W=128), dict(skips=[2]), dict(skips=[]), dict(in_channels_xyz=39), dict(in_channels_dir=15)):
        with pytest.raises(NotImplementedError, match="D=8, W=256"):
            sinnerf_amd.NeRF(**kw)
    with pytest.raises(ValueError, match="compute dtype"):
        sinnerf_amd.NeRF(compute_dtype="fp8")
    assert sinnerf_amd.NeRF(compute_dtype="fp16").compute_dtype == "fp16"    # round 6: an INFERENCE arithmetic (tests/test_fp16_gpu.py)


def test_no_torch_op_backend_in_the_package():
    """the product package holds no transcription of the reference's op sequence: `sinnerf_amd.generic` is gone, and neither the
    renderer nor the network imports torch.nn.functional / calls a torch op sequence for the MLP"""
    import sinnerf_amd
    import importlib
    with pytest.raises(ImportError):
        importlib.import_module("sinnerf_amd.generic")
    pkg = os.path.dirname(sinnerf_amd.__file__)
    for name in ("rendering.py", "nerf.py", "autograd.py"):
        src = open(os.path.join(pkg, name)).read()
        for needle in ("torch.nn.functional", "F.linear", "torch.cumprod", "torch.searchsorted", "FORCE_GENERIC", "from oracle", "import oracle"):
            assert needle not in src, (name, needle)


def test_embedding_verdict_is_cached_and_follows_the_bands():
    import sinnerf_amd
    from sinnerf_amd import rendering
    e = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    assert rendering._fused_embeddings(e)
    key0 = e[0]._sn_pow2_verdict[0]
    assert rendering._fused_embeddings(e) and e[0]._sn_pow2_verdict[0] == key0          # second call: the cached verdict
    assert not rendering._fused_embeddings([sinnerf_amd.Embedding(3, 10, logscale=False), e[1]])
    assert not rendering._fused_embeddings([sinnerf_amd.Embedding(3, 4), e[1]])          # right bands, wrong count
    assert not rendering._fused_embeddings([e[1], e[0]])
    e[0].freq_bands.mul_(2.0)                                                             # in place: _version moves, verdict re-taken
    assert not rendering._fused_embeddings(e)
    with pytest.raises(NotImplementedError, match="Embedding"):
        rendering._check_embeddings(e)
    # mutations that do NOT bump Tensor._version: .data swap, set_(), list-typed bands edited in place
    e2 = sinnerf_amd.Embedding(3, 10)
    assert rendering._is_pow2_bands(e2, 3, 10)
    e2.freq_bands.data = torch.linspace(1.0, 512.0, 10)
    assert not rendering._is_pow2_bands(e2, 3, 10)
    e3 = sinnerf_amd.Embedding(3, 4)
    assert rendering._is_pow2_bands(e3, 3, 4)
    e3.freq_bands.set_(torch.tensor([1.0, 2.0, 4.0, 9.0]))
    assert not rendering._is_pow2_bands(e3, 3, 4)

    class ListBands:
        in_channels, N_freqs = 3, 4
        freq_bands = [1.0, 2.0, 4.0, 8.0]
    lb = ListBands()
    assert rendering._is_pow2_bands(lb, 3, 4)
    lb.freq_bands[3] = 7.0
    assert not rendering._is_pow2_bands(lb, 3, 4)


def _lr_trace(sched, opt, n):
    out = []
    for _ in range(n):
        opt.step()
        sched.step()
        out.append((sched.last_epoch, opt.param_groups[0]["lr"]))
    return out


@pytest.mark.parametrize("advance", [0, 1, 2, 3, 4])
def test_rebuilt_scheduler_continues_the_same_schedule(advance):
    """MultiStepLR(new, ..., last_epoch=old.last_epoch) resumed one epoch ahead -- every milestone of the
    reference schedule (utils/__init__.py:34-36) fired a step early.  The rebuilt scheduler must sit at the old epoch with the
    old lr and then produce the very trajectory an un-rebuilt one does."""
    from sinnerf_amd.system import rebuild_scheduler
    def make():
        p = [torch.nn.Parameter(torch.zeros(3))]
        p[0].grad = torch.zeros(3)
        opt = torch.optim.Adam(p, lr=5e-4)
        return opt, torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3, 5], gamma=0.1)
    opt_a, sch_a = make()                          # never rebuilt
    opt_b, sch_b = make()                          # rebuilt after `advance` epochs
    _lr_trace(sch_a, opt_a, advance)
    _lr_trace(sch_b, opt_b, advance)
    p2 = [torch.nn.Parameter(torch.zeros(3))]
    p2[0].grad = torch.zeros(3)
    g = opt_b.param_groups[0]
    new = torch.optim.Adam(p2, lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"])
    new.param_groups[0]["initial_lr"] = g["initial_lr"]
    ns = rebuild_scheduler(sch_b, new)
    assert ns is not sch_b and ns.optimizer is new
    assert ns.last_epoch == sch_a.last_epoch == advance
    assert new.param_groups[0]["lr"] == opt_a.param_groups[0]["lr"]
    assert ns.get_last_lr() == sch_a.get_last_lr()
    assert _lr_trace(ns, new, 7) == _lr_trace(sch_a, opt_a, 7)


def test_checkpoint_key_filter(tmp_path):
    """utils/__init__.py:60-83: select one sub-module's entries, strip '<name>.', drop ignored prefixes; accepts a Lightning
    checkpoint, a bare state dict, or a path to either."""
    from sinnerf_amd.ckpt import extract_model_state_dict, load_ckpt
    t = lambda v: torch.full((2,), float(v))
    sd = {"nerf_coarse.sigma.weight": t(1), "nerf_coarse.rgb.0.bias": t(2), "nerf_fine.sigma.weight": t(3), "discriminator.conv.weight": t(4)}
    for ck in (sd, {"state_dict": sd, "epoch": 3}):
        got = extract_model_state_dict(ck, "nerf_coarse")
        assert set(got) == {"sigma.weight", "rgb.0.bias"} and got["sigma.weight"][0] == 1
        assert set(extract_model_state_dict(ck, "nerf_coarse", prefixes_to_ignore=["rgb"])) == {"sigma.weight"}
        assert set(extract_model_state_dict(ck, "nerf_fine")) == {"sigma.weight"}
        assert extract_model_state_dict(ck, "model") == {}
    path = os.path.join(tmp_path, "x.ckpt")
    torch.save({"state_dict": sd}, path)
    assert set(extract_model_state_dict(path, "nerf_fine")) == {"sigma.weight"}
    lin = torch.nn.Linear(2, 1)
    load_ckpt(lin, {"m.bias": torch.tensor([7.0])}, model_name="m")             # missing entries keep their values
    assert lin.bias.item() == 7.0
    with pytest.raises(RuntimeError):
        load_ckpt(lin, {"m.nope": torch.tensor([7.0])}, model_name="m")          # unknown entries fail in load_state_dict


def test_bench_line_keeps_every_named_config_and_stays_under_the_driver_tail():
    """BASELINE config 4's record used to be `dropped` from the one JSON line.  A full-size result (every record the
    N = 1 bench produces, with its prose) must compact to < 8 KB with the contract keys, `roofline`, `cpu_baseline` and the records of all
    named configs present."""
    import json
    import bench
    d = json.load(open(os.path.join(REPO, "profiles", "r06_run6_bench_full_fp32.json")))
    prose = "x" * 400
    res = dict(d)
    res["records"] = {k: dict(v, workload=prose, losses=prose, roofline=dict(v.get("roofline") or {}, kernel=prose, traffic_note=prose))
                      for k, v in d["records"].items()}
    for k in ("train_dp", "train_dp_graph", "train_dp_fp32", "train_step", "train_step_bf16", "train_step_bf16x3", "torch_eager_gpu_baseline"):
        if k in res:
            res[k] = dict(res[k], sample=prose, launch=prose, optimizer=prose)
    line = bench.compact_line(res)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    for k in ("bf16", "bf16x3", "config5_bf16", "train_cfg2_fp32", "train_cfg3_bf16", "train_cfg3_full_bf16", "train_cfg4_bf16"):
        assert k in line["records"], (k, line.get("dropped"))
    assert "torch_eager_gpu_baseline" in line and line["torch_eager_gpu_baseline"]["kind"] == "reference"


def test_committed_pmc_profile_is_only_used_for_the_kernel_sources_it_was_measured_on():
    """roofline.traffic falls back to profiles/pmc_traffic.json only when that profile carries THIS tree's kernel-source hash
    (the GPU box has no .git: a commit id could not be checked there); otherwise it is null with a note that says why."""
    import json
    import bench
    tj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha()
    got, note = bench.pmc_traffic(tj["points"], "fp32")                # no args: the committed-profile path
    if tj.get("kernel_sources_sha") == sha:
        assert got == tj["hbm_bytes"] and "NOT by this run" in note
    else:
        assert got is None and "other kernel sources" in note
    assert bench.pmc_traffic(12345, "fp32")[0] is None                 # another workload: no profile
