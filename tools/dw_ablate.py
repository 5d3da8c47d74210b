#!/usr/bin/env python3
"""Ablation of the dW kernel (sn_dw.hip): full / no in-loop DMA / no MFMA, variant 0 only, 256 tasks."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinnerf_amd import _lib
dev = torch.device("cuda:0")
P = 524288
G = torch.randn((8, P, 256), device=dev); X = torch.randn((8, P, 256), device=dev)
out = {}
for name, flag in (("full", 0), ("no_dma", 0x100), ("no_mfma", 0x200)):
    for nsplit in (32,):
        per = -(-P // nsplit // 16) * 16
        rows = []
        cp = torch.empty((8, nsplit, 256, 256), device=dev); bp = torch.empty((8, nsplit, 256), device=dev)
        for j in range(8):
            for s in range(nsplit):
                rows.append((G[j].data_ptr(), X[j].data_ptr(), cp[j, s].data_ptr(), bp[j, s].data_ptr(), s * per, min(P, (s + 1) * per),
                             256 | (256 << 32), 256 | ((0 | flag) << 32)))
        tasks = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
        f = lambda: _lib.check(_lib.lib.sn_dw_gemm(_lib.ptr(tasks), tasks.shape[0], None), "dw")
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        out[f"{name}_split{nsplit}"] = {"ms": ms, "tflops": 2 * 8 * 65536 * P / ms / 1e9, "GBps": 8 * 2048 * P / ms / 1e6}
    if name == "full":
        ref = G[0, :4096].double().t() @ X[0, :4096].double()
print(json.dumps(out, indent=1))
