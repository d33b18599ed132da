"""depthwise forward per MobileNetV2 layer shape at batch 16"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import *
lib = _lib.load(); N = 16; st = stream()
LAYERS = [("f1d", 32, 256, 1), ("f2d", 96, 256, 2), ("f3d", 144, 128, 1), ("f4d", 144, 128, 2), ("f5d", 192, 64, 1), ("f7d", 192, 64, 2),
          ("f8d", 384, 32, 1), ("f12d", 576, 32, 1), ("f14d", 576, 32, 2), ("f15d", 960, 16, 1)]
tot = 0
for name, Cc, H, s in LAYERS:
    Ho = H // s
    x = torch.randn(N, Cc, H, H, device="cuda"); w = torch.randn(Cc, 1, 3, 3, device="cuda"); y = torch.empty(N, Cc, Ho, Ho, device="cuda")
    cx = torch.zeros(Cc, SC_CST, device="cuda"); cx[:, 0] = 1
    xs = make_src(x, Cc, SRC_AFFINE, act=ACT_RELU6, cst=cx)
    rows = lib.sc_stat_rows(STAT_DW, N, Ho, Ho); stt = torch.empty(rows * Cc * 2, device="cuda")
    f = lambda: check(lib.sc_dwconv3x3_fwd(C.byref(xs), ptr(w), ptr(y), N, Cc, H, H, s, ptr(stt), st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    byt = 4.0 * N * Cc * (H * H + Ho * Ho)
    print(f"{name:5s} C={Cc:4d} {H:3d}^2 s{s}: {t*1e3:7.1f} us ({byt/t/1e6:6.0f} GB/s)")
    tot += t
print(f"sum {tot:.3f} ms")
