"""Does the bf16-state weight-gradient time depend on where acts / G / emb sit relative to each other (DRAM channel aliasing between
the A and B streams of a contraction)?  One arena, the three buffers carved at controlled gaps, slot stride = rows x 512 B with optional pad rows."""
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
arena = torch.empty(8 << 30, dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % (1 << 28)              # carve from a 256 MB-aligned address
def carve(off, shape, dtype):
    n = 1
    for s in shape: n *= s
    nb = n * torch.empty(0, dtype=dtype).element_size()
    return arena[base + off: base + off + nb].view(dtype).view(shape), off + nb
def run_case(pad_rows, gap):
    rows = P + pad_rows
    acts, off = carve(0, (10, rows, 256), torch.bfloat16)
    off = (off + gap + 255) // 256 * 256
    G, off = carve(off, (10, rows, 256), torch.bfloat16)
    off = (off + gap + 255) // 256 * 256
    emb, off = carve(off, (rows, 128), torch.float32)
    acts.normal_(); G.normal_(); emb.normal_()
    f = lambda: A._weight_grads(m, acts, emb, G, [True] * 24)
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("pad rows %5d  gap %9d B  (G - acts) mod 2^28 = %9d   dW total %.4f ms" % (pad_rows, gap, (G.data_ptr() - acts.data_ptr()) % (1 << 28), best), flush=True)
for pad in (0, 256, 768, 2048 + 256):
    for gap in (0, 4096, 65536 + 4096, (1 << 20) + 8192, (1 << 24) + 4096 * 3):
        run_case(pad, gap)
