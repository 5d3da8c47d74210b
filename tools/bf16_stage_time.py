"""time the three MLP stages of the bf16-state training step on the fine pass (524 288 points) -- used with SINNERF_HIP_LIB to
compare experimental builds"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import autograd as A, _lib
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
N = 4096
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::39][:N]).to(dev)
z = torch.sort(torch.rand((N, S), device=dev) * 4 + 2, -1)[0].contiguous()
g = torch.randn((N, S, 4), device=dev)
P = N * S
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r
ms_f, raw = timed(lambda: A._MLPFn.apply(m, rays, z, *m.raw_tensors()))
acts, emb, out = raw.grad_fn.saved_tensors
G = torch.zeros((10, acts.shape[1], 256), dtype=acts.dtype, device=dev); g_o = torch.empty((P, 4), device=dev)
ms_c, _ = timed(lambda: _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd("bf16")), 2, _lib.ptr(acts), _lib.ptr(out), _lib.ptr(g), P,
                                                                  acts.shape[1], _lib.ptr(G), _lib.ptr(g_o), None), "chain"))
ms_w, _ = timed(lambda: A._weight_grads(m, acts, emb, G, [True] * 24))
print("S=%d: fwd_train %.3f  chain %.3f  dW %.3f ms" % (S, ms_f, ms_c, ms_w))
