"""the step's small bookkeeping kernels, timed alone: BCE loss (+ dlogits), Adam, head forward / backward -- python tools/bench_small.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import *
lib = _lib.load(); st = stream()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, H, W = 16, 512, 512
z = torch.randn(N, 1, H, W, device="cuda"); t = (torch.rand(N, 1, H, W, device="cuda") > 0.9).float(); w = torch.rand(N, 1, H, W, device="cuda")
acc = torch.zeros(1, dtype=torch.float64, device="cuda"); dz = torch.empty_like(z)
print(f"bce loss + dlogits, {N}x{H}x{W}: {timeit(lambda: lib.sc_bce_logits_weighted(ptr(z), ptr(t), ptr(w), 1.0, z.numel(), ptr(acc), ptr(dz), None, st)):.1f} us")
print(f"bce dlogits only:            {timeit(lambda: lib.sc_bce_logits_weighted(ptr(z), ptr(t), ptr(w), 1.0, z.numel(), None, ptr(dz), None, st)):.1f} us")
n = 6629233
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4)); v.abs_()
print(f"adam, {n} parameters:       {timeit(lambda: lib.sc_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.03, 1.0, None, st)):.1f} us")
