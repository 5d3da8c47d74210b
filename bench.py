#!/usr/bin/env python3
"""bench.py -- rendered rays/s of the render_rays hot path on MI355X (BASELINE.json metric).

Workload (BASELINE configs[1]): nerf_synthetic/lego-shaped 400x400 frame = 160 000 synthetic pin-hole rays,
N_samples=64, N_importance=64, fp32, eval mode (perturb=0, noise_std=0, white_back=True), random-init
("teacher") NeRF weights.  One step = one full pass of the hot path (coarse MLP -> compositing -> sample_pdf ->
fine MLP -> compositing) over that frame, inputs resident in HBM.  N>1: every rank renders its own frame
(rays shard across ranks, no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0) following the driver contract, plus `roofline` (dominant kernel = fused MLP,
MFMA-bound) and `cpu_baseline` (the numpy oracle on the host cores, bounded sample, N=1 only).
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
FLOP_PER_POINT_SIGMA_ONLY = 2 * (63 * 256 + 3 * 256 * 256 + 319 * 256 + 3 * 256 * 256 + 256)
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}      # MI355X_MICROARCH.md: dense MFMA peaks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--hw", type=int, nargs=2, default=[400, 400])
    ap.add_argument("--n-importance", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")          # RCCL

    import sinnerf_amd
    from sinnerf_amd import rendering
    from oracle import oracle_np as O          # inputs generator + cpu_baseline leg only

    H, W = args.hw
    models, params = [], []
    for seed in (0, 1):
        p = O.init_params(seed, teacher=True)
        m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=args.dtype)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        models.append(m.to(dev).eval()); params.append(p)
    emb = [sinnerf_amd.Embedding(3, 10), sinnerf_amd.Embedding(3, 4)]
    rays_np = O.lego_rays(H, W, seed=rank)
    rays = torch.from_numpy(rays_np).to(dev)
    n_rays = rays.shape[0]
    NS, NI = 64, args.n_importance

    def step():
        with torch.no_grad():
            return rendering.render_rays(models, emb, rays, NS, False, 0, 0, NI, 1024 * 32 * 16, True)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    rendering.PROFILE = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    prof, rendering.PROFILE = rendering.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(v).all() for v in out.values())

    if rank == 0:
        total_rays = n_rays * world * args.steps
        value = total_rays / dt
        # dominant kernel: the fused MLP launch over the fine samples (2/3 of all points)
        fine = [(n, e0.elapsed_time(e1)) for (n, so, e0, e1) in prof if n == n_rays * (NS + NI)]
        coarse = [(n, e0.elapsed_time(e1)) for (n, so, e0, e1) in prof if n == n_rays * NS]
        ms_fine = float(np.mean([t for _, t in fine]))
        ms_coarse = float(np.mean([t for _, t in coarse]))
        flop_fine = FLOP_PER_POINT * n_rays * (NS + NI)
        achieved = flop_fine / (ms_fine * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes of the same launch from the rocprofv3 PMC passes of this command (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, tools/summarize_prof.py); PMC cannot be sampled from inside the process, so the committed summary
        # is quoted when it matches this workload, else null.
        traffic, traffic_note = None, "no PMC summary for this workload"
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
            if tj.get("points") == n_rays * (NS + NI) and args.dtype == "fp32":
                traffic = tj["hbm_bytes"]
                traffic_note = "bytes/launch from %s (algorithmic %d)" % (tj["source"], tj["algorithmic_bytes"])
        except Exception:
            pass
        res = {
            "metric": "rendered rays/sec (64+%d samples), lego %dx%d" % (NI, W, H),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "nerf_synthetic/lego-shaped %dx%d frame (%d rays/GPU), N_samples=64, N_importance=%d, "
                                   "eval render (perturb=0, noise_std=0, white_back), random-init teacher weights"
                                   % (W, H, n_rays, NI),
                       "rays_per_step_per_gpu": n_rays, "points_per_ray": NS + NS + NI, "parallelism": "rays sharded x%d, no collective" % world},
            "roofline": {"bound": "mfma", "kernel": "mlp_fwd_%s_kernel (fine pass, %d points/launch)" % ("f32" if args.dtype == "fp32" else "bf16", n_rays * (NS + NI)),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": traffic_note,
                         "flop_per_launch": flop_fine, "avg_launch_ms": ms_fine,
                         "coarse_launch_ms": ms_coarse,
                         "mlp_share_of_step": (ms_fine + ms_coarse) / (dt / args.steps * 1e3)},
            "roofline_rays_per_s_per_gpu": peak * 1e12 / (FLOP_PER_POINT * (NS + NS + NI)),
        }
        if world == 1:
            # secondary figure: one optimisation-shaped step (fwd+bwd of render_rays, 4096 rays, perturb=1, noise_std=1).
            # fp32: every kernel is MFMA-bound (fraction of the fp32 peak).  bf16 = mixed precision (bf16-operand forward,
            # chain and weight gradients over bf16 activations / gradients in HBM, ~26 KB per sample point: 5.6 written by
            # the forward, 10.25 moved by the chain, ~10.5 read by dW).
            try:
                for m in models:
                    m.train()
                tr = rays[:: n_rays // 4096][:4096].contiguous()
                tgt = torch.rand((4096, 3), device=dev)

                def tstep():
                    for m in models:
                        m.zero_grad(set_to_none=True)
                    r = rendering.render_rays(models, emb, tr, NS, False, 1.0, 1.0, NI, 32768, True)
                    (((r["rgb_fine"] - tgt) ** 2).mean() + ((r["rgb_coarse"] - tgt) ** 2).mean()).backward()
                tstep(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    tstep()
                torch.cuda.synchronize()
                tdt = (time.perf_counter() - t1) / 3
                pts = 4096 * (NS + NS + NI)
                res["train_step"] = {"rays": 4096, "ms": tdt * 1e3, "rays_per_s": 4096 / tdt}
                if args.dtype == "fp32":
                    tflop = 3489024 * pts / tdt / 1e12
                    res["train_step"].update({"bound": "mfma", "achieved_tflops": tflop, "frac_of_fp32_mfma_peak": tflop / peak})
                else:
                    tbs = 26000.0 * pts / tdt / 1e12
                    res["train_step"].update({"bound": "mixed (hbm / issue)", "approx_hbm_tb_per_s": tbs, "frac_of_8_tb_per_s": tbs / 8.0})
            except Exception as e:                      # noqa: BLE001
                res["train_step"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
            sample = rays_np[:: max(1, n_rays // args.cpu_rays)][:args.cpu_rays]
            t0 = time.perf_counter()
            O.render_rays(params, sample, NS, False, 0, 0, NI, 1024 * 32, True, False)
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": sample.shape[0] / cdt, "unit": "rays/s", "cores": ncpu, "kind": "port",
                                   "sample": "%d rays of the same frame, numpy/OpenBLAS oracle (oracle/oracle_np.py), "
                                             "%.1f s" % (sample.shape[0], cdt)}
        if world == 1 and not args.no_cpu_baseline:
            # the same algorithm as stock PyTorch-ROCm eager ops on this GPU (oracle/torch_ref.py, fp32, no_grad): the
            # "reference on the MI355X" figure SURVEY §8d asks for beside the CPU baseline.  Reported, never the target.
            try:
                from oracle import torch_ref as T
                tp = [{k: torch.from_numpy(v).to(dev) for k, v in p.items()} for p in params]
                er = rays[:: max(1, n_rays // 16384)][:16384].contiguous()
                with torch.no_grad():
                    T.render(tp, er[:1024], NS, NI, True)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(0, er.shape[0], 4096):
                        T.render(tp, er[i:i + 4096], NS, NI, True)
                    torch.cuda.synchronize()
                edt = time.perf_counter() - t1
                res["torch_eager_gpu_baseline"] = {"value": er.shape[0] / edt, "unit": "rays/s", "kind": "port",
                                                   "sample": "%d rays of the same frame in chunks of 4096, stock torch fp32 ops on the same MI355X" % er.shape[0],
                                                   "speedup_of_value": value / (er.shape[0] / edt)}
            except Exception as e:                      # noqa: BLE001
                res["torch_eager_gpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
