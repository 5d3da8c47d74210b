"""bf16-state weight gradients of one seeded state -> file (argv[1]); with two files: compare them bit for bit.
Used to check the generated narrow kernel against the compiler-scheduled one (SINNERF_DW_NARROW_COMPILER=1): same tasks, same
summation order, so every gradient must be identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    bad = 0
    for i, (x, y) in enumerate(zip(a, b)):
        same = torch.equal(x, y)
        if not same:
            bad += 1
            d = (x - y).abs().max().item()
            print("grad %d %s DIFFERS: max |d| %.3e, max |x| %.3e, finite %s" % (i, tuple(x.shape), d, x.abs().max().item(), bool(torch.isfinite(y).all())))
    print("compared %d gradients: %d differ" % (len(a), bad))
    sys.exit(1 if bad else 0)
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import autograd as A
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
out = []
for P in (256 * 3, 4096 * 16 + 256, 4096 * 128):      # a task of 3 chunks / odd chunk counts / the bench shape
    g = torch.Generator(device=dev).manual_seed(P)
    acts = torch.randn((10, P, 256), device=dev, generator=g).bfloat16()
    G = torch.randn((10, P, 256), device=dev, generator=g).bfloat16()
    emb = torch.randn((P, 128), device=dev, generator=g).bfloat16()      # SN_DTYPE_EMB_BF16 form: the 64-wide bf16 shapes
    out += [x.float().cpu() for x in A._weight_grads(m, acts, emb, G, [True] * 24)]
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
print("saved %d gradients" % len(out))
