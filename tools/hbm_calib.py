"""achievable HBM rates on this box with stock kernels: read-only (sum), write-only (fill), copy -- calibration for the
HBM-bound training kernels"""
import torch
d = torch.device("cuda:0")
n = 1 << 30                                     # 4 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=d).normal_()
y = torch.empty_like(x)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
b = n * 4
print("read  (sum)   %.2f TB/s" % (b / timed(lambda: x.sum()) / 1e12))
print("read  (amax)  %.2f TB/s" % (b / timed(lambda: x.view(-1, 4096).amax(1)) / 1e12))
print("write (fill)  %.2f TB/s" % (b / timed(lambda: y.fill_(1.0)) / 1e12))
print("copy  (r+w)   %.2f TB/s total" % (2 * b / timed(lambda: y.copy_(x)) / 1e12))
print("axpy  (2r+w)  %.2f TB/s total" % (3 * b / timed(lambda: torch.add(x, y, out=y)) / 1e12))
