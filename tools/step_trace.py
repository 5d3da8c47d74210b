"""five bf16 training steps of the 4096-ray lego patch through SinNeRFSystem.train_step (eager or graph=True via argv[1] == 'graph');
run under `rocprofv3 --kernel-trace` to see what one step is made of (tools/summarize_prof.py summarises the trace)"""
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
graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
for _ in range(6):
    sysm.train_step(batch, graph=graph)
torch.cuda.synchronize()
