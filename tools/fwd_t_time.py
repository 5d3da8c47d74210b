"""kernel-only timing of the bf16-state training forward (fine pass of a 4096-ray step: 524 288 points) through the C ABI --
used with SINNERF_HIP_LIB (ablation builds) and SINNERF_COMPILER_SCHEDULED=1 (the compiler-scheduled kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import _lib
dev = torch.device("cuda:0")
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
N, S = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 128
rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::39][:N]).to(dev)
z = torch.sort(torch.rand((N, S), device=dev) * 4 + 2, -1)[0].contiguous()
P = N * S
rows = -(-P // 256) * 256
out = torch.empty((N, S, 4), device=dev)
acts = torch.empty((10, rows, 256), dtype=torch.bfloat16, device=dev)
flag = _lib.SN_DTYPE_COMPILER_SCHEDULED if int(os.environ.get("SINNERF_COMPILER_SCHEDULED", "0")) else 0
emb16 = not flag and not int(os.environ.get("SINNERF_EMB_FP32", "0"))
emb = torch.empty((rows, 128), dtype=torch.bfloat16 if emb16 else torch.float32, device=dev)
flag |= _lib.SN_DTYPE_EMB_BF16 if emb16 else 0
def run():
    _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(m.packed()), _lib.SN_DTYPE_BF16_STATE | flag, _lib.ptr(rays), _lib.ptr(z), N, S,
                                             _lib.ptr(out), _lib.ptr(acts), _lib.ptr(emb), rows, _lib.stream_ptr()), "fwd")
for _ in range(3): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
bpp = 9 * 512 + 256 + 256 + (192 if emb16 else 512) + 16
print("fwd_train kernel %.4f ms  (%.2f TB/s of %d B/point)" % (best, bpp * P / best / 1e9, bpp))
