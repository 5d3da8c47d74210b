#!/usr/bin/env python3
"""bench.py -- rendered rays/s of the render_rays hot path on MI355X (BASELINE.json metric).

Headline workload (BASELINE configs[1]): nerf_synthetic/lego-shaped 400x400 frame = 160 000 synthetic pin-hole rays,
N_samples=64, N_importance=64, fp32 (the reference's own precision), eval mode (perturb=0, noise_std=0, white_back=True),
random-init ("teacher") NeRF weights.  One step = one full pass of the hot path (coarse MLP -> compositing -> sample_pdf ->
fine MLP -> compositing) over that frame, inputs resident in HBM.  N>1: every rank renders its own frame (rays shard across
ranks, no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0) following the driver contract, plus `roofline` (dominant kernel = fused MLP fine pass,
MFMA-bound, timed with HIP events on the launch stream) and `cpu_baseline` (the reference's op sequence as stock torch
ops on the host cores, bounded sample, N=1 only).  Secondary records, all measured OUTSIDE the timed region of the headline
number and each with its own roofline:
  records.bf16               the same frame with bf16-operand MFMAs (north_star's contraction precision, configs 3/5)
  records.config5_bf16/fp32  BASELINE configs[4] shape on one GPU: 800x800, 64+128 samples
  train_step / train_step_bf16   fwd+bwd of one 4096-ray patch batch (perturb=1, noise_std=1)
  train_dp                   the data-parallel training leg: per-rank 4096-ray patch -> backward -> ONE all-reduce of the
                             flat gradient buffer (RCCL when N>1) -> fused Adam; replicas asserted identical
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_POINT = 1186816            # SURVEY §8d / BASELINE.md §2 (un-padded MACs x2)
FLOP_PER_POINT_TRAIN = 3489024      # forward + backward
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}      # MI355X_MICROARCH.md: dense MFMA peaks
# compute_dtype="bf16x3" under autograd = forward, backward chain and weight gradients in the 3-term split on the bf16 MFMA (fp32-level
# values; training state stored as (hi, lo) pairs in the bytes of the fp32 state).  Its records are priced against the pipe they RUN on
# (VERDICT r5 "weak" #2): three bf16 MFMAs per product -> achieved = 3 x algorithmic against the 2.5 PF bf16 peak; the useful rate is kept
# beside it as `algorithmic_tflops` / `x_fp32_mfma_peak` (no `frac` may exceed 1: tests/test_bench_launcher_cpu.py).
TRAIN_PEAK = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}
MFMA_PER_PRODUCT = {"fp32": 1, "bf16": 1, "bf16x3": 3}


def mfma_frac_fields(dtype, algorithmic_tflops):
    """(achieved on the pipe, peak of that pipe, frac) + the extra fields of a bf16x3 record."""
    ach = MFMA_PER_PRODUCT[dtype] * algorithmic_tflops
    out = {"achieved_tflops": ach, "peak_tflops": TRAIN_PEAK[dtype], "frac_of_mfma_peak": ach / TRAIN_PEAK[dtype]}
    if dtype == "bf16x3":
        out.update(algorithmic_tflops=algorithmic_tflops, x_fp32_mfma_peak=algorithmic_tflops / PEAK_TFLOPS["fp32"],
                   mfma_per_product=3)
    elif dtype == "fp32":
        out["frac_of_fp32_mfma_peak"] = algorithmic_tflops / PEAK_TFLOPS["fp32"]
    return out


def build_models(O, dev, dtype, train=False):
    import sinnerf_amd
    models, params = [], []
    for seed in (0, 1):
        p = O.init_params(seed, teacher=True)
        m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        models.append((m.to(dev).train() if train else m.to(dev).eval())); params.append(p)
    return models, params


def time_render(models, emb, rays, NS, NI, steps, warmup, barrier=None):
    """`steps` eval renders of `rays`; returns (seconds for all steps, fine-launch ms, coarse-launch ms) with the MLP launches
    timed by HIP events on the launch stream (rendering.PROFILE)."""
    from sinnerf_amd import rendering
    n_rays = rays.shape[0]

    def step():
        with torch.no_grad():
            return rendering.render_rays(models, emb, rays, NS, False, 0, 0, NI, 1024 * 32 * 16, True)

    for _ in range(warmup):
        step()
    rendering.PROFILE = []
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    (barrier or torch.cuda.synchronize)()
    dt = time.perf_counter() - t0
    prof, rendering.PROFILE = rendering.PROFILE, None
    assert all(torch.isfinite(v).all() for v in out.values())
    fine = [e0.elapsed_time(e1) for (n, so, e0, e1) in prof if n == n_rays * (NS + NI)]
    coarse = [e0.elapsed_time(e1) for (n, so, e0, e1) in prof if n == n_rays * NS]
    return dt, float(np.mean(fine)), float(np.mean(coarse))


def roofline_record(dtype, n_rays, NS, NI, ms_fine, ms_coarse, ms_step, traffic=None, traffic_note=None):
    flop_fine = FLOP_PER_POINT * n_rays * (NS + NI)
    achieved = flop_fine / (ms_fine * 1e-3) / 1e12
    if dtype == "bf16x3":
        # fp32-level accuracy from THREE bf16 MFMAs per product (csrc/sn_mlp_fwd_bf16x3.hip): the matrix pipe executes 3x the
        # algorithmic FLOPs -- `achieved` / `frac` count those against the bf16 peak, `algorithmic_tflops` is the useful rate
        # (what the 157.3 TF fp32 MFMA peak bounds for the exact-fp32 kernel)
        return {"bound": "mfma", "kernel": "mlp_fwd_bf16x3_kernel (fine pass, %d points/launch)" % (n_rays * (NS + NI)),
                "achieved": 3 * achieved, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": 3 * achieved / PEAK_TFLOPS["bf16"],
                "algorithmic_tflops": achieved, "x_fp32_mfma_peak": achieved / PEAK_TFLOPS["fp32"], "traffic": traffic,
                "traffic_note": traffic_note or "no PMC summary for this workload", "flop_per_launch": flop_fine,
                "mfma_flop_per_launch": 3 * flop_fine, "avg_launch_ms": ms_fine, "coarse_launch_ms": ms_coarse,
                "mlp_share_of_step": (ms_fine + ms_coarse) / ms_step}
    peak = PEAK_TFLOPS[dtype]
    return {"bound": "mfma",
            "kernel": "mlp_fwd_%s_kernel (fine pass, %d points/launch)" % ({"fp32": "f32g", "fp16": "bf16_v3 [fp16 operands]"}.get(dtype, "bf16_v3"), n_rays * (NS + NI)),
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
            "traffic_note": traffic_note or "no PMC summary for this workload",
            "flop_per_launch": flop_fine, "avg_launch_ms": ms_fine, "coarse_launch_ms": ms_coarse,
            "mlp_share_of_step": (ms_fine + ms_coarse) / ms_step}


def kernel_sources_sha():
    """sha256 over the kernel sources as they are COMPILED -- csrc/*.hip, *.h, the generated *.inc instruction streams (build products of
    tools/gen_*.py, present wherever the library was built) and the Makefile: what a committed PMC profile was measured ON.  A profile stamped
    with a different hash describes other kernels and is refused.  (The GPU box has no .git: a commit id cannot be checked there; the generators'
    own text is not hashed -- a docstring edit changes no kernel.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "sinnerf_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inc"))
                   + [os.path.join(csrc, "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def live_pmc_traffic(args, n_points, timeout=120):
    """HBM bytes of the fine-pass launch, measured BY THIS RUN: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate passes,
    they do not fit one; --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a child `bench.py --steps 1 --no-extra` of the
    same workload; the longest mlp_fwd dispatch of each pass is the fine pass.  FETCH_SIZE is doubled (gfx950 tallies wide coalesced
    reads at half size), both are KiB.  Returns (bytes, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    tmp = tempfile.mkdtemp(prefix="sn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("WORLD_SIZE", None)
    try:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", cnt, "--pmc", cnt, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline", "--no-pmc", "--dtype", args.dtype,
                   "--hw", str(args.hw[0]), str(args.hw[1]), "--n-importance", str(args.n_importance), "--full-json", os.path.join(tmp, "child.json")]
            # own session: on a timeout the whole group goes -- rocprofv3 AND the python child it started (killing only
            # rocprofv3 can leave the grandchild rendering on the GPU under the records that follow)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                raise
            if rc != 0:
                raise subprocess.CalledProcessError(rc, cmd)
            best = (-1.0, None)
            for f in glob.glob(os.path.join(tmp, "**", cnt + "_counter_collection.csv"), recursive=True):
                per = {}
                for r in csv.DictReader(open(f)):
                    if "mlp_fwd" not in r["Kernel_Name"] or r["Counter_Name"] != cnt:
                        continue
                    d = per.setdefault(r["Dispatch_Id"], [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), 0.0])
                    d[1] += float(r["Counter_Value"])
                for dur, val in per.values():
                    if dur > best[0]:
                        best = (dur, val)
            if best[1] is None:
                return None, "no mlp_fwd dispatch in the %s pass" % cnt
            out[cnt] = best[1] * 1024.0
        total = 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]
        return total, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes over a 1-step child of this "
                       "command), fine-pass dispatch: 2 x %.0f B fetched (gfx950 correction) + %.0f B written; algorithmic %d B"
                       % (out["FETCH_SIZE"], out["WRITE_SIZE"], n_points * 20))
    except Exception as e:                  # noqa: BLE001
        return None, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(n_points, dtype, args=None):
    """`roofline.traffic`: HBM bytes of the fine-pass launch.  Measured live (live_pmc_traffic) unless --no-pmc; otherwise / on failure
    the committed profile is used ONLY IF it was measured on these kernel sources (kernel_sources_sha) -- a stale profile yields null."""
    why = "--no-pmc"
    if args is not None and not args.no_pmc:
        got, why = live_pmc_traffic(args, n_points)
        if got is not None:
            return got, why
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if tj.get("points") == n_points and dtype == "fp32":
            if tj.get("kernel_sources_sha") == kernel_sources_sha():
                return tj["hbm_bytes"], "bytes/launch from %s (git %s, kernel sources %s = this tree's), NOT by this run [%s] (algorithmic %d)" % (
                    tj["source"], tj.get("git_head", "unstamped"), tj["kernel_sources_sha"], why, tj["algorithmic_bytes"])
            return None, "no traffic: %s; the committed profile %s was measured on other kernel sources (%s, this tree: %s)" % (
                why, tj.get("source"), tj.get("kernel_sources_sha", "unstamped"), kernel_sources_sha())
    except Exception:                       # noqa: BLE001
        pass
    return None, "no traffic: %s; no committed profile for this workload" % why


def train_step_record(O, dev, dtype, rays, NS, NI, reps=10):
    """fwd+bwd of render_rays on a 4096-ray batch (perturb=1, noise_std=1, MSE coarse+fine), gradients only."""
    import sinnerf_amd
    from sinnerf_amd import rendering
    models, _ = build_models(O, dev, dtype, train=True)
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    n = rays.shape[0]
    tr = rays[:: max(1, n // 4096)][:4096].contiguous()
    tgt = torch.rand((tr.shape[0], 3), device=dev)

    def tstep():
        for m in models:
            m.zero_grad(set_to_none=True)
        r = rendering.render_rays(models, emb, tr, NS, False, 1.0, 1.0, NI, 32768, True)
        (((r["rgb_fine"] - tgt) ** 2).mean() + ((r["rgb_coarse"] - tgt) ** 2).mean()).backward()
    for _ in range(3):
        tstep()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        tstep()
    torch.cuda.synchronize()
    tdt = (time.perf_counter() - t1) / reps
    pts = tr.shape[0] * (NS + NS + NI)
    tflop = FLOP_PER_POINT_TRAIN * pts / tdt / 1e12
    return {"rays": tr.shape[0], "ms": tdt * 1e3, "rays_per_s": tr.shape[0] / tdt, "bound": "hbm" if dtype == "bf16" else "mfma",
            **({"roofline": train_hbm_roofline(tdt * 1e3, pts)} if dtype == "bf16" else {}), **mfma_frac_fields(dtype, tflop)}


# bf16 mixed-precision training keeps its state (activations, pre-activation gradients) in bf16 in HBM and every stage streams
# it once.  Per sample point: the forward writes 9 x 512 B activations + slot 9 (256 B softplus outputs + 256 B ReLU sign words)
# + 192 B embedded inputs (bf16 operands, SN_DTYPE_EMB_BF16; 512 B as fp32 until round 3) + 16 B output = 5 328 B; the chain reads
# the sign words, the 256 B softplus tile and 2 x 16 B and writes 9 x 512 B + 256 B + the 64 B head block of gradients + 16 B =
# 5 488 B; the weight-gradient contractions read 8 x 1 024 B (256 x 256 problems) + 3 328 B (the six narrow ones) = 11 520 B
# (DESIGN.md §3).  (Through round 3's first profiles this constant double-counted slot 9's sign-word half and the unwritten part
# of G[9]: 23 488 B; the PMC counters of the same launches see 22 860 B.)  The step is bound by HBM, not by the MFMA rate.
TRAIN_BF16_BYTES_PER_POINT = (9 * 512 + 256 + 256 + 192 + 16) + (256 + 256 + 32 + 9 * 512 + 256 + 64 + 16) + (8 * 1024 + 3328)
HBM_PEAK_TBS = 8.0                                  # MI355X_MICROARCH.md: 8 TB/s spec
HBM_ACHIEVABLE_TBS = 6.3                            # same guide: ~6.3 TB/s achievable; tools/hbm_calib.py on this pool: 6.2 read / 6.7 write / 5.2 copy


def train_hbm_roofline(ms_per_step, n_points):
    """`roofline` of the bf16 training step: algorithmic HBM bytes / step time against the HBM peak; `traffic` = the bytes
    the PMC counters saw (the profile profiles/train_pmc_latest.json points at: FETCH_SIZE x2 + WRITE_SIZE of the MLP stages of the bf16 step)."""
    traffic, note = None, "no PMC summary"
    try:
        ptr = json.load(open(os.path.join(REPO, "profiles", "train_pmc_latest.json")))     # pointer written by tools/save_round.py
        path = os.path.join(REPO, "profiles", ptr["file"])
        pj = json.load(open(path))
        k = pj["kernels"]
        per_pt = sum(v["bytes_per_point"] for name, v in k.items() if v.get("step", "bf16" if "bf16" in name else "fp32") == "bf16")
        if pj.get("kernel_sources_sha") == kernel_sources_sha():
            traffic = per_pt * n_points
            note = "PMC bytes/point of the bf16 forward, chain and weight-gradient launches x points (profiles/%s, git %s, kernel sources %s = this tree's; NOT by this run)" % (
                os.path.basename(path), pj.get("git_head", "unstamped"), pj["kernel_sources_sha"])
        else:
            note = "no traffic: profiles/%s was measured on other kernel sources (%s, this tree: %s)" % (
                os.path.basename(path), pj.get("kernel_sources_sha", "unstamped"), kernel_sources_sha())
    except Exception:                               # noqa: BLE001
        pass
    alg = TRAIN_BF16_BYTES_PER_POINT * n_points
    ach = alg / (ms_per_step * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": "bf16 training step: mlp_fwd_bf16_t_kernel + mlp_bwd_chain_bf16_t_kernel + dw_bf16_asm_kernel + dw_narrow_bf16_asm_kernel (coarse + fine)",
            "achieved": ach, "peak": HBM_PEAK_TBS, "unit": "TB/s", "frac": ach / HBM_PEAK_TBS,
            "achievable_peak": HBM_ACHIEVABLE_TBS, "frac_of_achievable": ach / HBM_ACHIEVABLE_TBS, "traffic": traffic,
            "traffic_note": note, "algorithmic_bytes_per_step": alg}


TRAIN_CFGS = {
    # BASELINE configs[1..3] training shapes (SURVEY §8d "Configs restated"): rays per render of the FOUR renders of one step
    # (models/sinnerf.py:304-307: rays, rays_full, rays_side, rays_proj), white_back of the dataset
    "train_cfg2": ("nerf_synthetic/lego 400x400 patch_size=64: 4 x 4096 rays/step", True),
    "train_cfg3": ("llff/room 504x378 patch 63x84 sW=sH=4: 4096 + 5292 + 5292 + 4096 rays/step, white_back=False", False),
    "train_cfg4": ("dtu scan 640x512 patch 56x70 sW=sH=8: 4096 + 3920 + 3920 + 4096 rays/step (per rank)", True),
}


def train_cfg_batch(O, dev, cfg, seed=0):
    """Synthetic SinNeRF patch batch of one config: the keys of models/sinnerf.py:277-299 that reach the hot path."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if cfg == "train_cfg2":
        f = 0.5 * 400 / np.tan(0.5 * 0.6911112)
        patch = lambda sd: O.patch_rays(400, 400, f, O._look_at_c2w(np.array([2.4 + 0.1 * sd, -2.2, 2.3])), 2.0, 6.0,
                                        30 + 7 * sd, 40 + 5 * sd, 64, 64, 5, 5)
        rnd = lambda sd: O.lego_rays(400, 400, seed=seed + sd)[:: 39][:4096]
        lo, hi = 2.0, 6.0
    elif cfg == "train_cfg3":
        patch = lambda sd: O.llff_patch_rays(seed + sd)
        rnd = lambda sd: O.llff_like_rays(4096, seed + sd)
        lo, hi = 1.2, 8.0
    else:
        patch = lambda sd: O.dtu_patch_rays(seed + sd)
        rnd = lambda sd: O.dtu_patch_rays(seed + 10 + sd, pw=64, ph=64, sx=7, sy=7)
        lo, hi = 2.125, 4.525
    rays, full, side, proj = rnd(0), patch(1), patch(2), rnd(3)
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = lambda *sh: torch.rand(*sh, generator=g).to(dev)
    return {"rays": t(rays), "rgbs": u(rays.shape[0], 3), "depth": lo + (hi - lo) * u(rays.shape[0]),
            "rays_full": t(full), "rgbs_full": u(full.shape[0], 3),
            "rays_side": t(side), "side_rgb": u(side.shape[0], 3),
            "rays_proj": t(proj), "depth_proj": lo + (hi - lo) * u(proj.shape[0])}


def train_cfg_record(O, dev, dtype, cfg, steps=3, warmup=2):
    """One optimisation step of the reference's real shape on a BASELINE config: the four renders of sinnerf.py:304-307
    (perturb=1, noise_std=1, MSE on `rays` and the full patch, SmoothL1 depth on `rays` and the projected rays with
    depth_weight=1, MSE stand-in for the unseen-view losses that stay on PyTorch), backward of all four, ONE flat all-reduce
    (no-op at world 1) and the fused Adam -- through SinNeRFSystem.train_step."""
    from sinnerf_amd.system import SinNeRFSystem
    what, white_back = TRAIN_CFGS[cfg]
    torch.manual_seed(7)
    sysm = SinNeRFSystem(N_importance=64, compute_dtype=dtype, perturb=1.0, noise_std=1.0, white_back=white_back,
                         depth_weight=1.0).to(dev)
    batch = train_cfg_batch(O, dev, cfg)
    n_rays = sum(batch[k].shape[0] for k in ("rays", "rays_full", "rays_side", "rays_proj"))
    for _ in range(warmup):
        out = sysm.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = sysm.train_step(batch)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out["loss"]).item()
    pts = n_rays * 192
    tflop = FLOP_PER_POINT_TRAIN * pts / step_s / 1e12
    rec = {"workload": what, "renders_per_step": 4, "rays_per_step": n_rays, "points_per_step": pts, "dtype": dtype,
           "losses": "MSE(rays) + MSE(full patch) + SL1 depth(rays) + SL1 depth(proj) + MSE(side patch)",
           "ms_per_step": step_s * 1e3, "train_rays_per_s": n_rays / step_s, **mfma_frac_fields(dtype, tflop),
           "loss": float(out["loss"].detach())}
    if dtype == "bf16":
        rec["roofline"] = train_hbm_roofline(step_s * 1e3, pts)
    else:
        rec["roofline"] = {"bound": "mfma", "kernel": ("fp32 training step: mlp_fwd_f32g_kernel<STORE> + mlp_bwd_chain_f32_kernel + dw_f32_asm_kernel + "
                                                       "dw_narrow_f32_kernel" if dtype == "fp32" else
                                                       "bf16x3 training step: mlp_fwd_bf16x3_kernel<STORE> + mlp_bwd_chain_bf16x3_kernel + dw_kernel "
                                                       "(3-term split, (hi, lo) state)") + " (4 renders, coarse + fine)",
                           "achieved": MFMA_PER_PRODUCT[dtype] * tflop, "peak": TRAIN_PEAK[dtype], "unit": "TFLOP/s",
                           "frac": MFMA_PER_PRODUCT[dtype] * tflop / TRAIN_PEAK[dtype], "traffic": None,
                           **({"algorithmic_tflops": tflop, "x_fp32_mfma_peak": tflop / PEAK_TFLOPS["fp32"]} if dtype == "bf16x3" else {})}
    return rec


class SetupFailed(RuntimeError):
    """a multi-rank record's collective-free setup failed on SOME rank: every rank skips the record (agreed by all-reduce)"""


def setup_agreed(setup, dev, world):
    """Run the collective-FREE part of a multi-rank record (models, batches: what can run out of memory) and let all ranks agree on
    whether it worked before any of them enters a collective: a rank that failed alone would otherwise leave the others blocked in
    the record's first all-reduce until the driver's timeout.  Exceptions BEHIND this point propagate: a failure inside
    the collective-bearing part is fatal for the whole job, with a traceback, instead of a hang."""
    obj, err = None, None
    try:
        obj = setup()
    except AssertionError:
        raise
    except Exception as e:                          # noqa: BLE001
        err = repr(e)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not bool(t.item()) and err is None:
            err = "setup failed on another rank"
    if err:
        raise SetupFailed(err)
    return obj


class VitStandIn(torch.nn.Module):
    """Bench-side stand-in for the DINO ViT-S/16 of models/extractor.py:20-24 (`torch.hub.load('facebookresearch/dino:main', 'dino_vits16')`:
    weights from the network, not runnable offline): the SAME architecture from stock torch.nn -- 16x16 patch embedding, cls token, 197
    learned positions, 12 pre-norm blocks of width 384 / 6 heads / MLP ratio 4, GELU -- with RANDOM weights.  It exists
    to put the ViT term's cost (two 224x224 forwards + their backward into the side render per step, sinnerf.py:332-339) into the config-3
    record; its values mean nothing.  `feature(x)` = VitExtractor.get_feature_from_input(x)[-1][0, 0, :] after SinNeRF.get_vit_feature's
    resize + ImageNet normalisation (sinnerf.py:162-169): the cls token of the LAST BLOCK's output.  Never imported by sinnerf_amd/."""

    class Block(torch.nn.Module):
        def __init__(self, dim=384, heads=6):
            super().__init__()
            self.norm1, self.norm2 = torch.nn.LayerNorm(dim, eps=1e-6), torch.nn.LayerNorm(dim, eps=1e-6)
            self.qkv, self.proj = torch.nn.Linear(dim, 3 * dim), torch.nn.Linear(dim, dim)
            self.fc1, self.fc2 = torch.nn.Linear(dim, 4 * dim), torch.nn.Linear(4 * dim, dim)
            self.heads = heads

        def forward(self, x):
            B, N, C = x.shape
            q, k, v = self.qkv(self.norm1(x)).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
            a = torch.nn.functional.scaled_dot_product_attention(q, k, v)
            x = x + self.proj(a.transpose(1, 2).reshape(B, N, C))
            return x + self.fc2(torch.nn.functional.gelu(self.fc1(self.norm2(x))))

    def __init__(self, dim=384, depth=12, heads=6):
        super().__init__()
        self.patch_embed = torch.nn.Conv2d(3, dim, 16, 16)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = torch.nn.Parameter(torch.randn(1, 197, dim) * 0.02)
        self.blocks = torch.nn.ModuleList([VitStandIn.Block(dim, heads) for _ in range(depth)])
        for p in self.parameters():
            p.requires_grad_(False)                              # the reference never optimises the extractor

    def feature(self, x):
        mean = torch.tensor([0.485, 0.456, 0.406], device=x.device).reshape(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=x.device).reshape(1, 3, 1, 1)
        x = (torch.nn.functional.interpolate(x, size=(224, 224)) - mean) / std
        t = self.patch_embed(x).flatten(2).transpose(1, 2)
        t = torch.cat([self.cls_token.expand(t.shape[0], -1, -1), t], 1) + self.pos_embed
        for b in self.blocks:
            t = b(t)
        return t[0, 0, :]


def train_cfg3_full_record(O, dev, dtype="bf16", steps=8, warmup=3):
    """BASELINE configs[2] AS NAMED -- "llff/room 504x378 patch 63x84 sW/sH=4 full SinNeRF losses, bf16" -- on what is runnable offline:
    the llff four-render step with the DINO-ViT term of sinnerf.py:332-339 through a stand-in of the same architecture with random weights
    (VitStandIn above, vit_weight 10 = README "Step 1") AND the ADVERSARIAL term through the reference's UNMODIFIED models/discriminator.py (staged into
    oracle/_ref by build(); sinnerf.py:143-145 with --patch_size unset as in the README's LLFF commands, hinge loss, dis_weight 0.01 =
    README "Step 2") and BOTH optimiser passes per batch (pytorch-lightning 0.10 calls training_step once per optimiser, sinnerf.py:271):
    pass 0 = four renders with gradients + -mean(D(side patch)) + flat all-reduce + Adam; pass 1 = one no-grad side render, hinge D loss,
    D backward, opt_d.  Discriminator and ViT stay on stock PyTorch-ROCm (north_star).  Reports the whole step with both terms, the step
    with the discriminator only (round 5's record), and the share spent on the HIP path (the same step with the MSE stand-in for the side
    loss + the no-grad side render, both timed here)."""
    from oracle import stage_ref
    from sinnerf_amd.system import SinNeRFSystem
    if not stage_ref.discriminator_available():
        return {"error": "oracle/_ref/models/discriminator.py not staged on this box (build() stages it where /root/reference exists)"}
    dmod = stage_ref.load_discriminator()
    what, white_back = TRAIN_CFGS["train_cfg3"]
    psx, psy = 63, 84

    def make(with_d):
        torch.manual_seed(7)
        sysm = SinNeRFSystem(N_importance=64, compute_dtype=dtype, perturb=1.0, noise_std=1.0, white_back=white_back, depth_weight=1.0,
                             dis_weight=0.01 if with_d else 0.0)
        if with_d:
            sysm.attach_discriminator(dmod.Discriminator(conditional=False, policy="color,cutout", imsize=-1), patch_hw=(psx, psy))
        sysm = sysm.to(dev)
        sysm.configure_optimizers()
        return sysm
    batch = train_cfg_batch(O, dev, "train_cfg3")
    batch["real_patch"] = torch.rand((1, 3, psx, psy), device=dev)
    n_rays = sum(batch[k].shape[0] for k in ("rays", "rays_full", "rays_side", "rays_proj"))

    def timed(fn, n, w):
        """median of n individually synchronised steps after w warm-up steps: the discriminator's convolutions go through MIOpen, whose
        first calls per shape search for an algorithm (a fresh box has no cache) -- a mean over the first few steps measures that search"""
        for _ in range(w):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out
    np.random.seed(0)
    full = make(True)
    t_dis, (out_g, out_d) = timed(lambda: full.train_step_adversarial(batch), max(steps, 12), max(warmup, 6))
    assert torch.isfinite(out_g["loss"]).item() and torch.isfinite(out_d["loss"]).item()
    # ... plus the ViT term: loss_vit = mse(f(side rgb_coarse), ref_) + mse(f(side rgb_fine), ref_) (sinnerf.py:332-339), ref_ = f(real patch)
    torch.manual_seed(11)
    vit = VitStandIn().to(dev).eval()
    vit_weight = 10.0
    with torch.no_grad():
        ref_ = vit.feature(batch["real_patch"])
    adv = full._generator_adv_loss

    def side_with_vit(results_side, b):
        patch = lambda k: results_side[k].reshape(1, psx, psy, 3).permute(0, 3, 1, 2)
        mse = torch.nn.functional.mse_loss
        l_vit = mse(vit.feature(patch("rgb_coarse")), ref_) + mse(vit.feature(patch("rgb_fine")), ref_)
        return adv(results_side, b) + vit_weight * l_vit
    full.side_loss = side_with_vit
    t_full, (out_g, out_d) = timed(lambda: full.train_step_adversarial(batch), max(steps, 12), max(warmup, 6))
    assert torch.isfinite(out_g["loss"]).item() and torch.isfinite(out_d["loss"]).item()
    vit_params = sum(p.numel() for p in vit.parameters())
    side = batch["rays_side"].reshape(-1, 8)

    def side_render():
        with torch.no_grad():
            return full(side)
    t_side, _ = timed(side_render, steps, 2)
    plain = make(False)
    t_plain, _ = timed(lambda: plain.train_step(batch), steps, warmup)
    d_params = sum(p.numel() for p in full.D.parameters())
    pts = n_rays * 192
    return {"workload": what + " + 10 x ViT-S/16 feature loss (stand-in: DINO's architecture, random weights, %d parameters) + hinge GAN through the "
                               "reference Discriminator (imsize=-1: %d parameters, logits (1,1,12,18)), both optimiser passes" % (vit_params, d_params),
            "dtype": dtype, "rays_per_step": n_rays, "ms_per_step": t_full * 1e3, "train_rays_per_s": n_rays / t_full,
            "ms_per_step_discriminator_only": t_dis * 1e3,
            "hip_path_ms": (t_plain + t_side) * 1e3, "hip_path_share": (t_plain + t_side) / t_full,
            "generator_pass_standin_ms": t_plain * 1e3, "side_render_nograd_ms": t_side * 1e3,
            "vit_ms": (t_full - t_dis) * 1e3, "discriminator_and_glue_ms": (t_dis - t_plain - t_side) * 1e3,
            "losses": "pass 0: MSE(rays) + MSE(full patch) + SL1 depth(rays, proj) + 10 x [mse(vit(side coarse), ref) + mse(vit(side fine), ref)] "
                      "+ 0.01 x hinge G loss; pass 1: hinge D loss on real patch / detached side render",
            "vit": "stand-in: ViT-S/16 of DINO's shape from stock torch.nn, random weights (the trained extractor needs the network)",
            "loss_g": float(out_g["loss"].detach()), "loss_d": float(out_d["loss"].detach()),
            "roofline": train_hbm_roofline(t_plain * 1e3, pts) if dtype == "bf16" else None}


def train_dp_leg(O, dev, dtype, rank, world, steps, warmup=2, graph=False):
    """The multi-GPU training path of SURVEY §8e: replicas (broadcast at start), every rank draws its OWN 4096-ray patch,
    fwd + loss + bwd, ONE all-reduce (mean) of the flat 1 191 688-float gradient buffer over RCCL, fused Adam on the flat
    parameter buffer (FlatAdam / sn_adam_step).  Returns per-rank step time (max over ranks), the all-reduce time from HIP
    events around the collective, and asserts the replicas are still bit-identical afterwards."""
    import torch.distributed as dist
    from sinnerf_amd.system import SinNeRFSystem
    def setup():
        torch.manual_seed(1234 + rank)                           # replicas start DIFFERENT; setup_distributed() fixes that
        sysm = SinNeRFSystem(N_importance=64, compute_dtype=dtype, perturb=1.0, noise_std=1.0, white_back=True).to(dev)
        rays = torch.from_numpy(O.lego_rays(400, 400, seed=100 + rank)[::39][:4096]).to(dev)      # this rank's patch
        return sysm, rays, {"rays": rays, "rgbs": torch.rand((rays.shape[0], 3), device=dev)}
    sysm, rays, batch = setup_agreed(setup, dev, world)
    flat = sysm.setup_distributed()
    for _ in range(warmup):
        sysm.train_step(batch, graph=graph)
    flat.profile = []
    if world > 1:
        dist.barrier(device_ids=[dev.index]) if dist.get_backend() == "nccl" else dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = sysm.train_step(batch, graph=graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ar_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in flat.profile]
    flat.profile = None
    chk = sysm.replica_checksum()
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    identical, n_seen = True, 1
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        identical = all(torch.equal(c, allc[0]) for c in allc)
        n_seen = dist.get_world_size()
    assert identical, "replicas diverged: the gradient all-reduce / optimizer step is not replica-consistent"
    assert torch.isfinite(out["loss"]).item()
    step_s = float(t.item()) / steps
    pts = 4096 * 192
    tflop = FLOP_PER_POINT_TRAIN * pts / step_s / 1e12
    return {"rays_per_rank_per_step": int(rays.shape[0]), "steps": steps, "dtype": dtype, "ms_per_step": step_s * 1e3,
            "train_rays_per_s_total": world * rays.shape[0] / step_s, "per_rank_tflops": tflop,
            "frac_of_mfma_peak": tflop / PEAK_TFLOPS[dtype],
            "all_reduce_us": float(np.mean(ar_us)) if ar_us else 0.0, "all_reduce_bytes": int(flat.flat.numel() * 4),
            "all_reduce_backend": (dist.get_backend() if world > 1 else "none (world 1)"), "rccl_version": rccl_version(),
            "n_ranks_seen": n_seen, "replicas_identical_after": steps + warmup,
            **({"roofline": train_hbm_roofline(step_s * 1e3, pts)} if dtype == "bf16" else {}),
            "optimizer": "FlatAdam (sn_adam_step, one launch)", "loss": float(out["loss"].detach()),
            "launch": "zero/forward/loss/backward replayed from ONE captured HIP graph; all-reduce + Adam eager" if graph
                      else "eager (every kernel launched from Python)"}


# ---- the ONE JSON line: the driver keeps the last 8 KB of stdout, so the line stays well below that -----------------------------
LINE_BUDGET = 7000
_KEEP_STR = {"unit", "dtype", "bound", "kind", "error", "scaling", "data", "metric", "all_reduce_backend", "rccl_version", "vit"}


def _sig(x, n=5):
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")) or x == 0.0:
        return x
    return float("%.*g" % (n, x))


_ROOFLINE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_tflops", "frac_of_achievable", "avg_launch_ms")
_DROP_IN_RECORDS = {"thread_calibration_rays_per_s", "flop_per_launch", "algorithmic_bytes_per_step", "replicas_identical_after",
                    "all_reduce_bytes", "optimizer", "launch", "losses", "peak_tflops", "host_cores", "roofline_rays_per_s_per_gpu",
                    "renders_per_step", "points_per_step", "per_rank_tflops", "rays_per_rank_per_step", "loss", "peak_tflops",
                    "frac_of_fp32_mfma_peak", "bound", "steps"}


def _slim(v, top=False, key=None):
    """numbers to 5 significant digits; inside secondary records prose strings go (the full record is in --full-json)"""
    if isinstance(v, dict):
        out = {}
        if key == "roofline" and not top:
            v = {k: x for k, x in v.items() if k in _ROOFLINE_KEEP and x is not None}
        for k, x in v.items():
            if isinstance(x, str) and not top and k not in _KEEP_STR:
                continue
            if k in _DROP_IN_RECORDS and not top and key != "roofline":
                continue
            out[k] = _slim(x, key=k)
        return out
    if isinstance(v, (list, tuple)):
        return [_slim(x) for x in v]
    return _sig(v)


def fracs_above_one(obj, path=""):
    """every `frac*` entry above 1 in a result tree: a fraction of a peak above 1 is priced against the wrong peak"""
    bad = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            if isinstance(v, (dict, list)):
                bad += fracs_above_one(v, path + "/" + str(k))
            elif isinstance(v, float) and str(k).startswith("frac") and v > 1.0:
                bad.append((path + "/" + str(k), v))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            bad += fracs_above_one(v, path + "/%d" % i)
    return bad


def compact_line(res):
    """The printed line: driver-contract keys + `roofline` + `cpu_baseline` verbatim (strings shortened), then the secondary
    records in priority order -- the ones the review reads first come first, and whatever would push the line past LINE_BUDGET is
    named in `dropped` instead of silently cut by the driver's 8 KB tail."""
    head = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config") if k in res}
    head = {k: (_sig(v) if not isinstance(v, dict) else {kk: _sig(vv) for kk, vv in v.items()}) for k, v in head.items()}
    rl = dict(res.get("roofline", {}))
    for k in ("flop_per_launch",):
        rl.pop(k, None)
    head["roofline"] = {k: (_sig(v) if not isinstance(v, str) else v[:160]) for k, v in rl.items()}
    if "cpu_baseline" in res:
        cb = {k: v for k, v in res["cpu_baseline"].items() if k != "thread_calibration_rays_per_s"}
        head["cpu_baseline"] = {k: (_sig(v) if not isinstance(v, str) else v[:200]) for k, v in cb.items()}
    assert not fracs_above_one(res), "a roofline fraction above 1 is priced against the wrong peak: %r" % (fracs_above_one(res),)
    rec = dict(res.get("records", {}))
    # priority: the other arithmetics of the headline frame, then every NAMED BASELINE config (configs[2..4]: the training step shapes in
    # bf16 -- config 3 also with the discriminator -- and the 800x800 frame), the GPU-side baseline, then the remaining precisions of the
    # same shapes; the per-stage / data-parallel legs that repeat information go last (train_dp_fp32 is the first to be dropped)
    order = ["headline_frame_sharded", "bf16", "fp32", "bf16x3", "fp16", "config5_bf16", "config5_sharded", "train_cfg4_dp",
             "train_cfg2_fp32", "train_cfg3_bf16", "train_cfg3_full_bf16", "train_cfg4_bf16", "train_cfg2_bf16", "train_cfg2_bf16x3"]
    late = ["train_cfg3_fp32", "train_cfg4_fp32", "config5_bf16x3", "config5_fp32"]
    prio = [("records", k) for k in order if k in rec]
    prio += [(None, k) for k in ("torch_eager_gpu_baseline", "train_dp", "train_step_bf16x3") if k in res]
    prio += [("records", k) for k in late if k in rec]
    prio += [("records", k) for k in rec if k not in order and k not in late]
    prio += [(None, k) for k in ("train_step_bf16", "train_step", "train_dp_graph", "train_dp_fp32") if k in res]
    out, dropped = dict(head), []
    out["records"] = {}
    for where, k in prio:
        val = _slim(rec[k] if where else res[k])
        (out["records"] if where else out)[k] = val
        if len(json.dumps(out)) > LINE_BUDGET:
            (out["records"] if where else out).pop(k)
            dropped.append(k)
    if not out["records"]:
        out.pop("records")
    if dropped:
        out["dropped"] = dropped
    for k, v in res.items():                     # small scalars added by callers (n_ranks_seen ...)
        if k not in out and not isinstance(v, (dict, list)) and len(json.dumps(out)) < LINE_BUDGET:
            out[k] = _sig(v)
    return out


def config5_sharded_record(O, dev, rank, world, barrier, dist, steps=2, warmup=1, hw=(800, 800), dtype="bf16", n_importance=128):
    """BASELINE configs[4] as the SHARDED workload it names (train.py:51-52): ONE lego 800x800 frame, 64+128
    samples, bf16 operands, its 640 000 rays partitioned contiguously over the ranks (parallel.shard_rays: 80 000 per rank at
    8 ranks), no collective on the data path; the rgb tiles are gathered to rank 0 (parallel.gather_rows) OUTSIDE the timed
    region.  Strong scaling: the frame is fixed, value = frame rays / max-over-ranks time."""
    import sinnerf_amd
    from sinnerf_amd import parallel
    H, W = hw

    def setup():
        frame = O.lego_rays(H, W, seed=0)
        lo, hi = parallel.shard_bounds(frame.shape[0], rank, world)
        mine = torch.from_numpy(np.ascontiguousarray(frame[lo:hi])).to(dev)
        models, _ = build_models(O, dev, dtype)
        return frame, lo, hi, mine, models, [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    frame, lo, hi, mine, models, emb = setup_agreed(setup, dev, world)
    NI = n_importance
    dt, ms_fine, ms_coarse = time_render(models, emb, mine, 64, NI, steps, warmup, barrier)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    with torch.no_grad():
        from sinnerf_amd import rendering
        rgb = rendering.render_rays(models, emb, mine, 64, False, 0, 0, NI, 32768, True)["rgb_fine"]
    full = parallel.gather_rows(rgb, frame.shape[0])                      # not timed
    dt = float(t.item())
    rec = None
    if rank == 0:
        assert full is not None and full.shape == (frame.shape[0], 3) and bool(torch.isfinite(full).all())
        rec = {"workload": "lego %dx%d frame (%d rays) sharded over %d ranks (%d rays on rank 0), 64+%d samples, %s" % (
                   W, H, frame.shape[0], world, hi - lo, NI, dtype),
               "value": frame.shape[0] * steps / dt, "unit": "rays/s", "ms_per_frame": dt / steps * 1e3, "dtype": dtype,
               "scaling": "strong", "rays_per_rank": hi - lo, "n_ranks": world, "gathered_rows_on_rank0": int(full.shape[0]),
               "roofline": roofline_record(dtype, hi - lo, 64, NI, ms_fine, ms_coarse, dt / steps * 1e3)}
    del mine, models
    torch.cuda.empty_cache()
    return rec


def train_cfg_dp_record(O, dev, dtype, cfg, rank, world, dist, steps=20, warmup=3):
    """BASELINE configs[3] as named: the dtu four-render step on EVERY rank (its own patches), then the ONE flat all-reduce and the
    fused Adam (SinNeRFSystem.train_step); per-step time = max over ranks, `all_reduce_us` from HIP events around the
    collective, replicas asserted identical afterwards."""
    from sinnerf_amd.system import SinNeRFSystem
    what, white_back = TRAIN_CFGS[cfg]

    def setup():
        torch.manual_seed(77 + rank)
        sysm = SinNeRFSystem(N_importance=64, compute_dtype=dtype, perturb=1.0, noise_std=1.0, white_back=white_back,
                             depth_weight=1.0).to(dev)
        return sysm, train_cfg_batch(O, dev, cfg, seed=31 * rank)
    sysm, batch = setup_agreed(setup, dev, world)
    flat = sysm.setup_distributed()
    n_rays = sum(batch[k].shape[0] for k in ("rays", "rays_full", "rays_side", "rays_proj"))
    for _ in range(warmup):
        out = sysm.train_step(batch)
    flat.profile = []
    if world > 1:
        dist.barrier(device_ids=[dev.index]) if dist.get_backend() == "nccl" else dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = sysm.train_step(batch)
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    ar_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in flat.profile]
    flat.profile = None
    chk = sysm.replica_checksum()
    identical = True
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        identical = all(torch.equal(c, allc[0]) for c in allc)
    assert identical, "replicas diverged in the %s data-parallel step" % cfg
    assert torch.isfinite(out["loss"]).item()
    step_s = float(t.item()) / steps
    pts = n_rays * 192
    rec = {"workload": what + " -- on each of %d ranks, one flat all-reduce per step" % world, "dtype": dtype, "n_ranks": world,
           "renders_per_step": 4, "rays_per_step_per_rank": n_rays, "ms_per_step": step_s * 1e3,
           "train_rays_per_s_total": world * n_rays / step_s, "all_reduce_us": float(np.mean(ar_us)) if ar_us else 0.0,
           "all_reduce_bytes": int(flat.flat.numel() * 4), "all_reduce_backend": dist.get_backend() if world > 1 else "none (world 1)",
           "replicas_identical_after": steps + warmup, "scaling": "weak"}
    rec["roofline"] = train_hbm_roofline(step_s * 1e3, pts) if dtype == "bf16" else {
        "bound": "mfma", "achieved": FLOP_PER_POINT_TRAIN * pts / step_s / 1e12, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
        "frac": FLOP_PER_POINT_TRAIN * pts / step_s / 1e12 / PEAK_TFLOPS["fp32"], "traffic": None}
    del sysm
    torch.cuda.empty_cache()
    return rec


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, script=None, timeout=None):
    """Self-launch: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts N copies of itself, one rank per
    process (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set as torchrun would), waits for all of
    them and returns the worst exit code.  Rank 0 inherits stdout (its ONE JSON line is the job's); the other ranks' stdout
    goes to stderr.  The torchrun form (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) bypasses
    this: WORLD_SIZE is then already set."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, deadline = 0, (time.time() + timeout if timeout else None)
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or (deadline and time.time() > deadline):
                break                                   # one rank died (or time is up): do not leave the others in a rendezvous
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()                           # exact PIDs we started, never a pattern
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:                           # noqa: BLE001
                p.kill()
            rc = rc or (p.returncode if p.returncode is not None else 1)
    return rc


def launcher_selftest(args, rank, world):
    """`--selftest-launcher`: the rendezvous half of the bench without a GPU (what tests/test_bench_launcher_cpu.py runs with
    gloo): every rank joins the process group the way main() does, contributes to one all-reduce and one MAX-reduce of a
    time, rank 0 prints the JSON line."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(args.dist_backend)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    mx = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "sum_of_ranks_plus_1": float(t.item()), "max_time": float(mx.item()),
                          "all_reduce_backend": dist.get_backend() if world > 1 else "none (world 1)",
                          "self_launched": os.environ.get("SN_BENCH_SELF_LAUNCHED") == "1"}))
    if world > 1:
        dist.destroy_process_group()


def rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                   # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--hw", type=int, nargs=2, default=[400, 400])
    ap.add_argument("--n-importance", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline number + roofline only (used by the rocprof passes)")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the two rocprofv3 PMC passes that measure roofline.traffic live "
                                                            "(used by those passes themselves and by profiling runs of this script)")
    ap.add_argument("--cpu-rays", type=int, default=16384)
    ap.add_argument("--dist-backend", default="nccl", help="developer option: 'gloo' lets N ranks share one GPU for testing")
    ap.add_argument("--selftest-launcher", action="store_true", help="rendezvous + all-reduce only, no GPU work (CPU test)")
    ap.add_argument("--full-json", default=None, help="also write the un-shortened result (every record, every note) to this file; "
                                                       "default gpurun_out/bench_full_<dtype>_n<N>.json when gpurun_out/ can be created")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: be our own launcher (one process per GPU); under torchrun WORLD_SIZE is set
        if not args.selftest_launcher and args.dist_backend == "nccl":
            assert torch.cuda.is_available() and torch.cuda.device_count() >= args.gpus, \
                "--gpus %d needs %d visible devices (found %d); RCCL cannot put two ranks on one device -- use " \
                "--dist-backend gloo to exercise the N-rank path on fewer devices" % (
                    args.gpus, args.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}: the two must agree"
    if args.selftest_launcher:
        return launcher_selftest(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    n_dev = torch.cuda.device_count()
    if args.dist_backend != "nccl":
        local = local % n_dev
    else:
        # fail fast and legibly (torchrun path too): RCCL cannot put two ranks on one device, and a LOCAL_RANK beyond the visible
        # devices would otherwise surface as an opaque HIP error inside the first collective
        assert n_dev >= world and local < n_dev, (
            "rank %d: --gpus %d with backend nccl (RCCL) needs %d visible devices, found %d (LOCAL_RANK=%d; HIP_VISIBLE_DEVICES=%r) -- "
            "use --dist-backend gloo to exercise the N-rank path on fewer devices"
            % (rank, world, world, n_dev, local, os.environ.get("HIP_VISIBLE_DEVICES")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool: RCCL fails with hipIpcGetMemHandle otherwise
        # who is here, BEFORE the first collective (stderr: stdout carries rank 0's one JSON line) -- if the rendezvous or the first
        # all-reduce hangs, the log says which ranks arrived, on which device, with which RCCL
        print("[bench rank %d/%d] pid %d device cuda:%d (%s) of %d visible, backend %s, rccl %s, master %s:%s"
              % (rank, world, os.getpid(), local, torch.cuda.get_device_name(local), n_dev, args.dist_backend, rccl_version(),
                 os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")), file=sys.stderr, flush=True)
        dist.init_process_group(args.dist_backend, timeout=datetime.timedelta(minutes=10))          # "nccl" = RCCL
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                                     # the first collective: its own line in the log, not a record's
        assert int(seen.item()) == world, "first all-reduce saw %d ranks, expected %d" % (int(seen.item()), world)
        print("[bench rank %d/%d] first all-reduce ok: %d ranks" % (rank, world, int(seen.item())), file=sys.stderr, flush=True)

    import sinnerf_amd
    from oracle import oracle_np as O          # inputs generator + cpu_baseline leg only

    H, W = args.hw
    models, params = build_models(O, dev, args.dtype)
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    rays_np = O.lego_rays(H, W, seed=rank)
    rays = torch.from_numpy(rays_np).to(dev)
    n_rays = rays.shape[0]
    NS, NI = 64, args.n_importance

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local]) if args.dist_backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    dt, ms_fine, ms_coarse = time_render(models, emb, rays, NS, NI, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    res = None
    if rank == 0:
        total_rays = n_rays * world * args.steps
        value = total_rays / dt
        traffic, traffic_note = pmc_traffic(n_rays * (NS + NI), args.dtype, args if world == 1 else None)
        res = {
            "metric": "rendered rays/sec (64+%d samples), lego %dx%d" % (NI, W, H),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "nerf_synthetic/lego-shaped %dx%d frame (%d rays/GPU), N_samples=64, N_importance=%d, "
                                   "eval render (perturb=0, noise_std=0, white_back), random-init teacher weights"
                                   % (W, H, n_rays, NI),
                       "rays_per_step_per_gpu": n_rays, "points_per_ray": NS + NS + NI,
                       "parallelism": "rays sharded x%d, no collective" % world},
            "roofline": roofline_record(args.dtype, n_rays, NS, NI, ms_fine, ms_coarse, dt / args.steps * 1e3, traffic, traffic_note),
            "roofline_rays_per_s_per_gpu": PEAK_TFLOPS[args.dtype] * 1e12 / (FLOP_PER_POINT * (NS + NS + NI)),
        }

    # ---- the data-parallel training leg (every rank takes part; it is the only place a collective runs) ----------------
    if not args.no_extra:
        try:
            # (a 3.5 ms step: 5 warm-up + >= 20 timed steps -- a 2 + 5 run measures 7 % of ramp-up, tools/dp_leg_time.py; the
            # all-reduce time is the mean over those >= 20 steps)
            dp_steps = max(20, min(args.steps, 100))
            leg = train_dp_leg(O, dev, "bf16", rank, world, steps=dp_steps, warmup=5)
            leg_graph = train_dp_leg(O, dev, "bf16", rank, world, steps=dp_steps, warmup=5, graph=True) if world == 1 else None
            leg32 = train_dp_leg(O, dev, "fp32", rank, world, steps=3) if world == 1 else None
        except AssertionError:
            raise
        except SetupFailed as e:                    # agreed by every rank before any collective of the leg
            leg, leg_graph, leg32 = {"error": repr(e)}, None, None
        except Exception as e:                      # noqa: BLE001
            if world > 1:                           # inside the collective-bearing part: the other ranks are blocked in it -- fail the job, loudly
                raise
            leg, leg_graph, leg32 = {"error": repr(e)}, None, None
        if rank == 0:
            res["train_dp"] = leg
            if leg_graph is not None:
                res["train_dp_graph"] = leg_graph
            if leg32 is not None:
                res["train_dp_fp32"] = leg32

    if world > 1 and not args.no_extra:
        # ---- the named multi-GPU configs (BASELINE configs[3], configs[4]) as the sharded workloads they are: every rank takes part
        recs = {}
        if rank == 0:                                # the headline is measured: keep it legible even if a record below takes the job down
            print("[bench] headline before the multi-rank records: " + json.dumps({k: res[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step")}),
                  file=sys.stderr, flush=True)
        # STRONG scaling of the headline workload (VERDICT r5 "weak" #8): ONE frame of the headline shape (400x400, 64+64, headline dtype)
        # partitioned over the ranks -- 20 000 rays per GPU at 8 -- beside the weak-scaling headline, so that one N-GPU run yields both curves
        for key, fn in (("headline_frame_sharded", lambda: config5_sharded_record(O, dev, rank, world, barrier, dist, steps=max(3, args.steps // 4),
                                                                                   warmup=1, hw=(H, W), dtype=args.dtype, n_importance=NI)),
                        ("config5_sharded", lambda: config5_sharded_record(O, dev, rank, world, barrier, dist,
                                                                            hw=(800, 800) if (H, W) == (400, 400) else (2 * H, 2 * W))),
                        ("train_cfg4_dp", lambda: train_cfg_dp_record(O, dev, "bf16", "train_cfg4", rank, world, dist))):
            try:
                recs[key] = fn()
            except SetupFailed as e:                # every rank agreed to skip this record; anything else (a failure between
                recs[key] = {"error": repr(e)}      # collectives) propagates and ends the job with a traceback instead of a hang
        if rank == 0:
            res["records"] = recs

    if rank == 0 and world == 1 and not args.no_extra:
        # ---- secondary records: other precisions / configs, each with its own roofline (never the headline `value`) ----
        records = {}
        try:
            other = "bf16" if args.dtype == "fp32" else "fp32"
            mo, _ = build_models(O, dev, other)
            d2, f2, c2 = time_render(mo, emb, rays, NS, NI, 3, 1)
            records[other] = {"value": n_rays * 3 / d2, "unit": "rays/s", "ms_per_step": d2 / 3 * 1e3,
                              "workload": "same frame, %s-operand MFMAs (fp32 accumulate)" % other,
                              "roofline": roofline_record(other, n_rays, NS, NI, f2, c2, d2 / 3 * 1e3),
                              "roofline_rays_per_s_per_gpu": PEAK_TFLOPS[other] * 1e12 / (FLOP_PER_POINT * (NS + NS + NI))}
            # the same frame at FP32-LEVEL accuracy on the bf16 matrix cores (3-term hi/lo split; held to the fp32 parity bars by
            # tests/test_bf16x3_gpu.py): a secondary record -- the headline `value` stays the exact-fp32 kernel
            m3, _ = build_models(O, dev, "bf16x3")
            d3, f3, c3 = time_render(m3, emb, rays, NS, NI, 3, 1)
            with torch.no_grad():
                r3 = sinnerf_amd.render_rays(m3, emb, rays, NS, False, 0, 0, NI, 1 << 19, True)["rgb_fine"]
                r32 = sinnerf_amd.render_rays(models if args.dtype == "fp32" else mo, emb, rays, NS, False, 0, 0, NI, 1 << 19, True)["rgb_fine"]
            records["bf16x3"] = {"value": n_rays * 3 / d3, "unit": "rays/s", "ms_per_step": d3 / 3 * 1e3, "dtype": "bf16x3",
                                 "workload": "same frame, fp32-level accuracy from 3 bf16 MFMAs per product (hi/lo split of weights and activations)",
                                 "speedup_vs_fp32_kernel": (n_rays * 3 / d3) / (value if args.dtype == "fp32" else records["fp32"]["value"]),
                                 "max_abs_rgb_diff_vs_fp32_kernel": float((r3 - r32).abs().max()),
                                 "psnr_vs_fp32_kernel_dB": float(-10 * torch.log10(torch.mean((r3 - r32) ** 2).clamp_min(1e-20))),
                                 "roofline": roofline_record("bf16x3", n_rays, NS, NI, f3, c3, d3 / 3 * 1e3)}
            del m3, r3
            # fp16 operands on the bf16 instruction streams (SN_DTYPE_F16, round 6): the bf16 rate, 8 more operand bits
            mh, _ = build_models(O, dev, "fp16")
            dh, fh, ch = time_render(mh, emb, rays, NS, NI, 3, 1)
            with torch.no_grad():
                rh = sinnerf_amd.render_rays(mh, emb, rays, NS, False, 0, 0, NI, 1 << 19, True)["rgb_fine"]
            records["fp16"] = {"value": n_rays * 3 / dh, "unit": "rays/s", "ms_per_step": dh / 3 * 1e3, "dtype": "fp16",
                               "workload": "same frame, fp16-operand MFMAs (fp32 accumulate): the bf16 kernels' instruction streams with v_cvt_pk_f16_f32",
                               "max_abs_rgb_diff_vs_fp32_kernel": float((rh - r32).abs().max()),
                               "psnr_vs_fp32_kernel_dB": float(-10 * torch.log10(torch.mean((rh - r32) ** 2).clamp_min(1e-20))),
                               "roofline": roofline_record("fp16", n_rays, NS, NI, fh, ch, dh / 3 * 1e3)}
            del mh, rh, r32
            if (H, W, NI) == (400, 400, 64):
                big = torch.from_numpy(O.lego_rays(800, 800, seed=0)).to(dev)          # BASELINE configs[4] shape, one GPU
                for dt_name, k in (("bf16", 2), ("bf16x3", 1), ("fp32", 1)):
                    mb, _ = build_models(O, dev, dt_name)
                    d5, f5, c5 = time_render(mb, emb, big, 64, 128, k, 1)
                    records["config5_" + dt_name] = {
                        "value": big.shape[0] * k / d5, "unit": "rays/s", "ms_per_step": d5 / k * 1e3,
                        "workload": "lego 800x800 frame (640 000 rays), 64+128 samples, %s, one GPU" % dt_name,
                        "roofline": roofline_record(dt_name, big.shape[0], 64, 128, f5, c5, d5 / k * 1e3),
                        "roofline_rays_per_s_per_gpu": PEAK_TFLOPS.get(dt_name, PEAK_TFLOPS["bf16"] / 3) * 1e12 / (FLOP_PER_POINT * 256)}
                del big
        except Exception as e:                      # noqa: BLE001
            records["error"] = repr(e)
        res["records"] = records
        for key, dt_name in (("train_step", "fp32"), ("train_step_bf16", "bf16"), ("train_step_bf16x3", "bf16x3")):
            try:
                res[key] = train_step_record(O, dev, dt_name, rays, NS, NI)
            except Exception as e:                  # noqa: BLE001
                res[key] = {"error": repr(e)}
        # the reference's real optimisation step (four renders + depth loss) on the BASELINE training shapes
        for cfg in TRAIN_CFGS:
            for dt_name in ("bf16", "fp32") + (("bf16x3",) if cfg == "train_cfg2" else ()):
                try:
                    records[cfg + "_" + dt_name] = train_cfg_record(O, dev, dt_name, cfg, steps=8 if dt_name == "bf16" else 3, warmup=3 if dt_name == "bf16" else 2)
                except Exception as e:              # noqa: BLE001
                    records[cfg + "_" + dt_name] = {"error": repr(e)}
                torch.cuda.empty_cache()
        try:
            records["train_cfg3_full_bf16"] = train_cfg3_full_record(O, dev, "bf16")
        except Exception as e:                      # noqa: BLE001
            records["train_cfg3_full_bf16"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the reference's op sequence (models/rendering.py:126-335 + models/nerf.py) as stock torch ops on
        # the reference's own runtime -- torch CPU, all host cores -- on a bounded sample of the same frame (oracle/torch_ref.py;
        # /root/reference itself cannot travel to the GPU box).  The numpy restatement is kept as a secondary figure.
        ncpu = os.cpu_count() or 1
        sample = np.ascontiguousarray(rays_np[:: max(1, n_rays // args.cpu_rays)][:args.cpu_rays])
        from oracle import torch_ref as T
        from oracle import stage_ref
        srays = torch.from_numpy(sample)
        if stage_ref.available():
            # the reference's OWN modules (models/rendering.py:126 render_rays, models/nerf.py NeRF / Embedding), staged
            # byte-for-byte into the git-ignored oracle/_ref/ by build() where /root/reference exists (oracle/stage_ref.py)
            ref_rendering, _ = stage_ref.load()
            ref_models, ref_emb = stage_ref.build_reference_models(params)
            kind = "reference"

            def cpu_render(r, white_back=True):
                return ref_rendering.render_rays(ref_models, ref_emb, r, NS, False, 0, 0, NI, 1024 * 32, white_back)
        else:
            tp = [{k: torch.from_numpy(v) for k, v in p.items()} for p in params]
            kind = "port"

            def cpu_render(r, white_back=True):
                return T.render(tp, r, NS, NI, white_back)
        # thread count: torch's intra-op pool with ALL cores of a 256-core host is 10x SLOWER than with a few dozen on this
        # op sequence (measured: 26 rays/s at 256 threads) -- so the count is calibrated on a 256-ray probe and the best one
        # is used and reported as `cores`
        cal = {}
        with torch.no_grad():
            for nt in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
                torch.set_num_threads(nt)
                cpu_render(srays[:64])                                                   # warm the pool at this size
                t0 = time.perf_counter()
                cpu_render(srays[:256])
                cal[nt] = 256 / (time.perf_counter() - t0)
                if time.perf_counter() - t0 > 8.0:
                    break
            nthreads = max(cal, key=cal.get)
            torch.set_num_threads(nthreads)
            nb = min(sample.shape[0], max(1024, int(cal[nthreads] * 15) // 1024 * 1024))  # ~15 s of CPU work
            cpu_render(srays[:256])
            t0 = time.perf_counter()
            for i in range(0, nb, 1024):                                                 # eval.py-style ray chunks
                cpu_render(srays[i:i + 1024])
            cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": nb / cdt, "unit": "rays/s", "cores": nthreads, "host_cores": ncpu, "kind": kind,
                               "sample": "%d rays of the same frame in chunks of 1024 (eval.py:84-115), %s, torch CPU with %d "
                                         "intra-op threads, %.1f s" % (
                                             nb, "the UNMODIFIED reference render_rays + NeRF modules (oracle/_ref, staged by "
                                             "oracle/stage_ref.py)" if kind == "reference" else
                                             "stock torch CPU ops (oracle/torch_ref.py = the reference's op sequence; oracle/_ref "
                                             "not staged on this box)", nthreads, cdt),
                               "thread_calibration_rays_per_s": {str(k): v for k, v in cal.items()}}
        try:                                       # BASELINE configs[0] as named: ONE 1024-ray chunk of a 504x378 llff-shaped frame, 64+64, CPU
            lr = torch.from_numpy(O.llff_like_rays(1024, seed=0))
            with torch.no_grad():
                cpu_render(lr[:256], False)
                t0 = time.perf_counter()
                cpu_render(lr, False)
            res["cpu_baseline"]["config1_llff_1024_rays_per_s"] = 1024 / (time.perf_counter() - t0)
        except Exception as e:                      # noqa: BLE001
            res["cpu_baseline"]["config1_llff_1024_rays_per_s"] = repr(e)
        if kind == "reference":                                                          # the port beside it, same sample
            try:
                tp = [{k: torch.from_numpy(v) for k, v in p.items()} for p in params]
                with torch.no_grad():
                    T.render(tp, srays[:256], NS, NI, True)
                    t0 = time.perf_counter()
                    for i in range(0, min(nb, 2048), 1024):
                        T.render(tp, srays[i:i + 1024], NS, NI, True)
                res["cpu_baseline"]["port_rays_per_s"] = min(nb, 2048) / (time.perf_counter() - t0)
            except Exception as e:                  # noqa: BLE001
                res["cpu_baseline"]["port_rays_per_s"] = repr(e)
        try:
            ns = sample[:1024]
            t0 = time.perf_counter()
            O.render_rays(params, ns, NS, False, 0, 0, NI, 1024 * 32, True, False)
            ndt = time.perf_counter() - t0
            res["cpu_baseline"]["numpy_oracle_rays_per_s"] = ns.shape[0] / ndt
        except Exception as e:                      # noqa: BLE001
            res["cpu_baseline"]["numpy_oracle_rays_per_s"] = repr(e)
        # the reference ITSELF run eagerly on this GPU through PyTorch-ROCm: the "reference on the MI355X" figure SURVEY §8d asks for
        # beside the CPU baseline, and the only measurable stand-in for north_star's "reference PyTorch-CUDA rays/sec".  The staged,
        # unmodified render_rays + NeRF modules with eval.py's OWN chunking: batched_inference slices rays by chunk = 1024*32*16
        # (eval.py:92 -- the whole 160 000-ray frame is one call) and render_rays chunks POINTS by the same value (rendering.py:196).
        # Reported, never the target.  (oracle/torch_ref.py, the port, only when oracle/_ref is not staged.)
        try:
            if stage_ref.available():
                gm, gemb = stage_ref.build_reference_models(params)
                gm = [m.to(dev) for m in gm]
                gkind, gchunk = "reference", 1024 * 32 * 16

                def gpu_render(r):
                    outs = [ref_rendering.render_rays(gm, gemb, r[i:i + gchunk], NS, False, 0, 0, NI, gchunk, True, test_time=False)
                            for i in range(0, r.shape[0], gchunk)]                 # eval.py:94-110 (test_time=False, eval.py:107)
                    return outs[-1]["rgb_fine"]
                er = rays
            else:
                tg = [{k: torch.from_numpy(v).to(dev) for k, v in p.items()} for p in params]
                gkind, gchunk = "port", 4096

                def gpu_render(r):
                    for i in range(0, r.shape[0], gchunk):
                        o = T.render(tg, r[i:i + gchunk], NS, NI, True)
                    return o["rgb_fine"]
                er = rays[:: max(1, n_rays // 16384)][:16384].contiguous()
            with torch.no_grad():
                gpu_render(er[:4096])
                gpu_render(er)                                                       # warm: allocator pools at the full size
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                reps = 2
                for _ in range(reps):
                    last = gpu_render(er)
                torch.cuda.synchronize()
            edt = (time.perf_counter() - t1) / reps
            assert bool(torch.isfinite(last).all())
            ev = er.shape[0] / edt
            res["torch_eager_gpu_baseline"] = {
                "value": ev, "unit": "rays/s", "kind": gkind,
                "sample": ("the UNMODIFIED reference render_rays + NeRF (oracle/_ref) on this MI355X through stock PyTorch-ROCm fp32 ops, the whole %d-ray "
                           "frame per call with eval.py's chunk = %d (eval.py:92, rendering.py:196), test_time=False (eval.py:107), %d timed frames" % (er.shape[0], gchunk, reps))
                          if gkind == "reference" else "%d rays in chunks of 4096, oracle/torch_ref.py (port) on this MI355X" % er.shape[0],
                "speedup_of_value": res["value"] / ev}
            for k in ("bf16", "bf16x3", "fp32"):
                if k in res.get("records", {}) and "value" in res["records"][k]:
                    res["torch_eager_gpu_baseline"]["speedup_of_%s_record" % k] = res["records"][k]["value"] / ev
            del last
            torch.cuda.empty_cache()
        except Exception as e:                      # noqa: BLE001
            res["torch_eager_gpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        full = args.full_json or os.path.join(REPO, "gpurun_out", "bench_full_%s_n%d.json" % (args.dtype, world))
        try:
            os.makedirs(os.path.dirname(full), exist_ok=True)
            json.dump(res, open(full, "w"), indent=1)
        except OSError:
            pass
        print(json.dumps(compact_line(res)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
