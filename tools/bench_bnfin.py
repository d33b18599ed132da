"""what a BatchNorm finalize launch costs in a dependent chain: python tools/bench_bnfin.py
(sc_bn_finalize / sc_bn_bwd_finalize on typical (channels, partial rows) of the 512^2 batch-16 step, against a one-thread kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import SC_CST, check, ptr, stream
lib = _lib.load(); dev = "cuda"


def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


step = torch.zeros(1, dtype=torch.int64, device=dev); lr = torch.ones(1, device=dev); hp = torch.zeros(4, device=dev)
print(f"one-thread kernel (sc_adam_prepare): {timeit(lambda: check(lib.sc_adam_prepare(ptr(step), ptr(lr), 0.9, 0.999, ptr(hp), stream()))):.2f} us per launch")
for C_, rows in ((384, 512), (96, 2048), (960, 128), (64, 512), (24, 2048), (32, 8192), (16, 32768)):
    stats = torch.rand(rows, C_, 2, device=dev)
    g, b, rm, rv = torch.ones(C_, device=dev), torch.zeros(C_, device=dev), torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    cst = torch.empty(C_, SC_CST, device=dev); scratch = torch.empty(64 * 2 * C_, dtype=torch.float64, device=dev)
    t = timeit(lambda: check(lib.sc_bn_finalize(ptr(stats), rows, float(rows * 32), ptr(g), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, 1, ptr(cst), C_,
                                                ptr(scratch), None, stream())))
    sums = torch.rand(min(rows, 4096), C_, 2, device=dev, dtype=torch.float64)
    dg, db, cb = torch.empty(C_, device=dev), torch.empty(C_, device=dev), torch.empty(C_, SC_CST, device=dev)
    t2 = timeit(lambda: check(lib.sc_bn_bwd_finalize(ptr(sums), sums.shape[0], float(rows * 32), ptr(cst), ptr(dg), ptr(db), ptr(cb), C_, stream())))
    print(f"C {C_:4d} rows {rows:6d}: sc_bn_finalize {t:6.2f} us   sc_bn_bwd_finalize ({sums.shape[0]} rows) {t2:6.2f} us")
