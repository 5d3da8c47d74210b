"""time the training forward of the fine pass (524288 points) -- used with SINNERF_HIP_LIB to compare experimental builds"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import autograd as A
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::39][:4096]).to(dev)
z = torch.sort(torch.rand((4096, 128), device=dev) * 4 + 2, -1)[0].contiguous()
def run():
    return A._MLPFn.apply(m, rays, z, *m.raw_tensors())
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print("fwd_train_ms %.3f" % (e0.elapsed_time(e1) / 5))
