"""depthwise backward: fused kernel vs the three separate ones, per MobileNetV2 layer shape at batch 16 (ms, GB/s of algorithmic bytes)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import *
lib = _lib.load()
N = 16
LAYERS = [("f1d", 32, 256, 1), ("f2d", 96, 256, 2), ("f3d", 144, 128, 1), ("f4d", 144, 128, 2), ("f5d", 192, 64, 1), ("f7d", 192, 64, 2),
          ("f8d", 384, 32, 1), ("f12d", 576, 32, 1), ("f14d", 576, 32, 2), ("f15d", 960, 16, 1)]
st = stream()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0, 0, 0]
for name, Cc, H, s in LAYERS:
    Ho = H // s
    x = torch.randn(N, Cc, H, H, device="cuda"); g = torch.randn(N, Cc, Ho, Ho, device="cuda"); y = torch.randn(N, Cc, Ho, Ho, device="cuda")
    w = torch.randn(Cc, 1, 3, 3, device="cuda")
    cd = torch.zeros(Cc, SC_CST, device="cuda"); cd[:, 0] = 1; cd[:, 2] = 1
    cx = torch.zeros(Cc, SC_CST, device="cuda"); cx[:, 0] = 1; cx[:, 3] = 1
    ds = make_src(g, Cc, SRC_BNBWD, act=ACT_RELU6, cst=cd, aux=y); xs = make_src(x, Cc, SRC_AFFINE, act=ACT_RELU6, cst=cx)
    dx = torch.empty_like(x); acc = torch.zeros(Cc * 9, dtype=torch.float64, device="cuda")
    rows = lib.sc_stat_rows(STAT_DW, N, H, H); sums = torch.empty(rows * Cc * 2, dtype=torch.float64, device="cuda")
    brows = lib.sc_stat_rows(STAT_BNBWD, N, H, H); bs = torch.empty(brows * Cc * 2, dtype=torch.float64, device="cuda")
    tf = timeit(lambda: check(lib.sc_dwconv3x3_bwd_fused(C.byref(ds), C.byref(xs), ptr(w), ptr(dx), ptr(acc), ptr(sums), N, Cc, H, H, s, st)))
    td = timeit(lambda: check(lib.sc_dwconv3x3_dgrad(C.byref(ds), ptr(w), ptr(dx), 0, N, Cc, H, H, s, st)))
    tw = timeit(lambda: check(lib.sc_dwconv3x3_wgrad(C.byref(ds), C.byref(xs), ptr(acc), N, Cc, H, H, s, st)))
    tr = timeit(lambda: check(lib.sc_bn_bwd_reduce(ptr(dx), ptr(x), ptr(cx), ACT_RELU6, ptr(bs), N, Cc, H * H, None, None, st)))
    byt = 4.0 * N * Cc * (2 * Ho * Ho + 2 * H * H)
    print(f"{name:5s} C={Cc:4d} {H:3d}^2 s{s}: fused {tf*1e3:7.1f} us ({byt/tf/1e6:7.0f} GB/s)   dgrad {td*1e3:6.1f} + wgrad {tw*1e3:6.1f} + bn-reduce {tr*1e3:6.1f} = {(td+tw+tr)*1e3:7.1f} us")
    tot[0] += tf; tot[1] += td + tw; tot[2] += tr
print(f"sum over these shapes: fused {tot[0]:.3f} ms, dgrad+wgrad {tot[1]:.3f} ms, bn-reduce {tot[2]:.3f} ms")
