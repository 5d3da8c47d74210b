"""time the weight-gradient launch of the fine pass; mode 'hot' aliases every row to row 0 (lda = ldb = 0: same instruction
stream, no HBM traffic) to separate memory effects from issue effects"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import _lib
from tests.helpers import dw_tasks
dev = torch.device("cuda:0")
P = 4096 * 128
acts = torch.randn((10, P, 256), device=dev); G = torch.randn((10, P, 256), device=dev); emb = torch.randn((P, 128), device=dev)
rows, outs = dw_tasks(acts, emb, G)
def timed(rows):
    tasks = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    def run(): _lib.check(_lib.lib.sn_dw_gemm(_lib.ptr(tasks), tasks.shape[0], None), "dw")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
print("tasks", len(rows), "normal %.3f ms" % timed(rows))
hot = [r[:6] + (0, r[7]) for r in rows]
print("hot (ld=0) %.3f ms" % timed(hot))
rows16, _ = dw_tasks(acts, emb, G, bf16=True)
print("bf16 operands %.3f ms" % timed(rows16))
print("bf16 hot (ld=0) %.3f ms" % timed([r[:6] + (0, r[7]) for r in rows16]))
by16 = {}
for r in rows16: by16.setdefault((r[7] >> 32) & 0xff, []).append(r)
for v, rs in sorted(by16.items()): print("bf16 variant", v, "tasks", len(rs), "alone %.3f ms" % timed(rs))
by_var = {}
for r in rows: by_var.setdefault(r[7] >> 32, []).append(r)
for v, rs in sorted(by_var.items()):
    print("variant", v, "tasks", len(rs), "alone %.3f ms" % timed(rs), " points/task", rs[0][5] - rs[0][4])

acts16, G16 = acts.bfloat16(), G.bfloat16()
rows_s, _ = dw_tasks(acts16, emb, G16, bf16=True)
print("bf16 state %.3f ms" % timed(rows_s))
print("bf16 state hot (ld=0) %.3f ms" % timed([r[:6] + (0, r[7]) for r in rows_s]))
bys = {}
for r in rows_s: bys.setdefault((r[7] >> 32) & 0xff, []).append(r)
for v, rs in sorted(bys.items()): print("bf16 state variant", v, "tasks", len(rs), "alone %.3f ms" % timed(rs), "points/task", rs[0][5] - rs[0][4])
