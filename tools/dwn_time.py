"""kernel times of the bf16-state weight-gradient launches (fine pass) from HIP events around sn_weight_grads, and per kernel from a
short rocprof-free loop: prints total ms (asm + narrow + finish)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import autograd as A
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
P = 4096 * 128
acts = torch.randn((10, P, 256), device=dev).bfloat16(); G = torch.randn((10, P, 256), device=dev).bfloat16(); emb = torch.randn((P, 128), device=dev).bfloat16()
def run(): return A._weight_grads(m, acts, emb, G, [True] * 24)
for _ in range(3): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("dW total %.4f ms" % best)
