"""per-variant cost of the bf16x3 weight-gradient contractions (run_task modes 4..7 of dw_kernel, flags 0x100 | 0x400; random bits
stand in for the (hi, lo) state -- timing only) through the raw
sn_dw_gemm entry: each variant's tasks alone -> cycles-equivalent cost per point and workgroup, normalised to variant 0 = 512
(the K-split table COST_X3 of csrc/sn_dw.hip)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from sinnerf_amd import _lib
from tests.helpers import dw_tasks
dev = torch.device("cuda:0")
P = 4096 * 128
acts = torch.randn((10, P, 256), device=dev); G = torch.randn((10, P, 256), device=dev); emb = torch.randn((P, 128), device=dev)
rows16, _ = dw_tasks(acts, emb, G, bf16=True)
rows = [r[:7] + (r[7] | (0x400 << 32),) for r in rows16]
def timed(rows):
    tasks = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    def run(): _lib.check(_lib.lib.sn_dw_gemm(_lib.ptr(tasks), tasks.shape[0], None), "dw")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
print("all tasks", len(rows), "%.3f ms" % timed(rows), " (bf16 x1 on the same state: %.3f ms)" % timed(rows16))
by = {}
for r in rows: by.setdefault((r[7] >> 32) & 0xff, []).append(r)
cost = {}
for v, rs in sorted(by.items()):
    n_prob = len({(r[0], r[1]) for r in rs})
    t = timed(rs)
    pts = rs[0][5] - rs[0][4]
    cost[v] = t / pts                      # ms per point of a K-range on one CU (every task of the variant runs concurrently: <= 256 tasks)
    print("variant", v, "problems", n_prob, "tasks", len(rs), "points/task", pts, "alone %.3f ms" % t)
print("COST_X3 =", [round(512 * cost[v] / cost[0]) for v in sorted(cost)])
