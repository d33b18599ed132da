"""1x1 weight gradient on the encoder's pointwise shapes (B=16): us per call incl. the reduce kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, wgrad_mfma
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU6, SC_CST, make_src
N = 16
SH = [("f1p", 32, 16, 256), ("f2e", 16, 96, 256), ("f2p", 96, 24, 128), ("f3e", 24, 144, 128), ("f3p", 144, 24, 128), ("f4p", 144, 32, 64), ("f5e", 32, 192, 64), ("f5p", 192, 32, 64),
      ("f7p", 192, 64, 32), ("f8e", 64, 384, 32), ("f8p", 384, 64, 32), ("f11p", 384, 96, 32), ("f12e", 96, 576, 32), ("f12p", 576, 96, 32),
      ("f14p", 576, 160, 16), ("f15e", 160, 960, 16), ("f15p", 960, 160, 16), ("f17p", 960, 320, 16), ("f18", 320, 1280, 16)]
tot = 0.0
for name, cin, cout, H in SH:
    x = torch.randn(N, cin, H, H, device=DEV); g = torch.randn(N, cout, H, H, device=DEV); y = torch.randn(N, cout, H, H, device=DEV)
    cst = torch.rand(cout, SC_CST, device=DEV); cstx = torch.rand(cin, SC_CST, device=DEV)
    dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU6, cst=cst, aux=y)
    src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU6, cst=cstx)
    fn = lambda: wgrad_mfma(dys, [src], N, H, H, cout, cin, 1)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    tot += t
    print(f"{name:5s} {cin:5d}->{cout:5d} {H:3d}^2  {t:7.1f} us  chk {float(fn().double().sum()):.6e}")
print(f"total {tot:.1f} us")
