"""bench.train_dp_leg at the bench's default length (2 warm-up + 5 steps) and at 5 + 20 / 5 + 50: how much of the short run is ramp-up"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle_np as O
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for graph in (False, True):
    for warm, steps in ((5, 20), (5, 50)):
        r = bench.train_dp_leg(O, dev, "bf16", 0, 1, steps=steps, warmup=warm, graph=graph)
        print("graph=%d warmup %d steps %2d: %.4f ms/step" % (graph, warm, steps, r["ms_per_step"]), flush=True)
