"""sc_bn_bwd_reduce at the step's large shapes (batch 16): us per launch and TB/s of its two input tensors.   usage: python tools/bench_bnbwd.py"""
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from starcop_amd import _lib
from starcop_amd._lib import ACT_RELU, SC_CST, STAT_BNBWD, check, ptr, stream
lib = _lib.load()
dev = torch.device("cuda:0")
N = 16
for name, C_, S in (("decoder.blocks.4", 16, 512), ("decoder.blocks.3", 32, 256), ("features.1 dw", 32, 256), ("features.3 dw", 144, 128), ("decoder.blocks.2", 64, 128), ("decoder.blocks.1", 128, 64)):
    g, y = torch.randn(N, C_, S, S, device=dev), torch.randn(N, C_, S, S, device=dev)
    cst = torch.rand(C_, SC_CST, device=dev) + 0.5
    rows = lib.sc_stat_rows(STAT_BNBWD, N, S, S)
    sums = torch.empty(rows * C_ * 2, dtype=torch.float64, device=dev)
    amax, aamax = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    f = lambda: check(lib.sc_bn_bwd_reduce(ptr(g), ptr(y), ptr(cst), ACT_RELU, ptr(sums), N, C_, S * S, ptr(amax), ptr(aamax), stream()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = 2 * N * C_ * S * S * 4 / 1e6
    print(f"{name:18s} {C_:4d} x {S}^2  {mb:5.0f} MB  {rows:5d} rows  {us:7.1f} us  {mb / us:5.2f} TB/s")
