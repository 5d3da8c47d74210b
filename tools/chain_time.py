"""time the backward chain of the fine pass (524288 points) -- used with SINNERF_HIP_LIB to compare experimental builds"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import autograd as A, _lib
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::39][:4096]).to(dev)
z = torch.sort(torch.rand((4096, 128), device=dev) * 4 + 2, -1)[0].contiguous()
raw = A._MLPFn.apply(m, rays, z, *m.raw_tensors())
acts, embt, outt = raw.grad_fn.saved_tensors
P = 4096 * 128
g = torch.randn((4096, 128, 4), device=dev)
G = torch.zeros((10, acts.shape[1], 256), device=dev); g_o = torch.empty((P, 4), device=dev)
def run():
    _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd()), 0, _lib.ptr(acts), _lib.ptr(outt), _lib.ptr(g), P, acts.shape[1],
                                              _lib.ptr(G), _lib.ptr(g_o), None), "chain")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
with torch.no_grad():
    sinnerf_amd.rendering._mlp(m, rays, z, False); torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(5): sinnerf_amd.rendering._mlp(m, rays, z, False)
    f1.record(); torch.cuda.synchronize()
print("chain_ms %.3f   (fwd_infer_ms %.3f on this box)" % (e0.elapsed_time(e1) / 5, f0.elapsed_time(f1) / 5))
