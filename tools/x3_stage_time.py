#!/usr/bin/env python3
"""bf16x3 training stages through the C ABI on the fine-pass shape (4096 rays x 128 samples = 524 288 points): the generated-stream
kernels against the compiler-scheduled ones they replace (SN_DTYPE_COMPILER_SCHEDULED), bit identity of everything they write, and HIP-event
times.  usage: x3_stage_time.py [reps]      (run under rocprofv3 --kernel-trace --pmc ... for cycles / MFMA-busy)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O                      # noqa: E402  (input generator only)
import sinnerf_amd                                     # noqa: E402
from sinnerf_amd import _lib                           # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_rays, S = 4096, 128
rays = torch.from_numpy(O.lego_rays(400, 400, seed=0)[::39][:n_rays]).to(dev)
z = torch.sort(torch.rand((n_rays, S), device=dev) * 4 + 2, -1)[0].contiguous()
P = n_rays * S
rows = -(-P // 128) * 128
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev)
X3 = _lib.SN_DTYPE_BF16X3
g_raw = torch.randn((P, 4), device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, flag in (("generated", 0), ("compiler-scheduled", _lib.SN_DTYPE_COMPILER_SCHEDULED)):
    out = torch.zeros((n_rays, S, 4), device=dev)
    acts = torch.zeros((10, rows, 256), device=dev)
    emb = torch.zeros((rows, 128), device=dev)
    G = torch.zeros((10, rows, 256), device=dev)
    g_o = torch.zeros((P, 4), device=dev)
    code = m.kernel_dtype(X3) | flag
    fwd = lambda: _lib.check(_lib.lib.sn_mlp_forward_train(_lib.ptr(m.packed()), code, _lib.ptr(rays), _lib.ptr(z), n_rays, S, _lib.ptr(out),
                                                           _lib.ptr(acts), _lib.ptr(emb), rows, _lib.stream_ptr()), "fwd")
    chain = lambda: _lib.check(_lib.lib.sn_mlp_backward_chain(_lib.ptr(m.packed_bwd("bf16x3")), code, _lib.ptr(acts), _lib.ptr(out), _lib.ptr(g_raw), P, rows,
                                                              _lib.ptr(G), _lib.ptr(g_o), _lib.stream_ptr()), "chain")
    t_f, t_c = timed(fwd), timed(chain)
    res[name] = (out, acts, emb, G, g_o)
    print("%-20s forward %.3f ms   chain %.3f ms   (%d points)" % (name, t_f, t_c, P))
a, b = res["generated"], res["compiler-scheduled"]
for nm, x, y in zip(("out", "acts", "emb", "G", "g_out"), a, b):
    same = torch.equal(x.view(torch.int32), y.view(torch.int32))
    print("  %-5s bit-identical: %s%s" % (nm, same, "" if same else "  (%d words differ)" % int((x.view(torch.int32) != y.view(torch.int32)).sum())))
