"""steady-state time of the 4096-ray bf16 training step (SinNeRFSystem.train_step, eager): 5 warm-up steps, best of 4 x 20 steps.
Used with SINNERF_HIP_LIB (variant builds) and SINNERF_EMB_FP32 / SINNERF_COMPILER_SCHEDULED for A/B runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
from sinnerf_amd.system import SinNeRFSystem
dev = torch.device("cuda:0")
torch.manual_seed(0)
sysm = SinNeRFSystem(N_importance=64, compute_dtype="bf16", perturb=1.0, noise_std=1.0, white_back=True).to(dev)
sysm.setup_distributed()
rays = torch.from_numpy(O.lego_rays(400, 400, seed=100)[::39][:4096]).to(dev)
batch = {"rays": rays, "rgbs": torch.rand((4096, 3), device=dev)}
modes = (False, True) if "--graph" in sys.argv else (False,)
for graph in modes:
    for _ in range(5): sysm.train_step(batch, graph=graph)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): sysm.train_step(batch, graph=graph)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%s step %.4f ms" % ("graph" if graph else "eager", best), flush=True)
