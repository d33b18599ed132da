"""1x1 conv: 128-pixel-tile kernel vs split-K kernel on the encoder's pointwise shapes (B=16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU6, SC_CST, make_src
from starcop_amd.network import _pick_cot
N = 16
SH = [("f1p", 32, 16, 256), ("f2e", 16, 96, 256), ("f2p", 96, 24, 128), ("f3e", 24, 144, 128), ("f4p", 144, 32, 64), ("f5e", 32, 192, 64), ("f5p", 192, 32, 64),
      ("f7p", 192, 64, 32), ("f8e", 64, 384, 32), ("f8p", 384, 64, 32), ("f11p", 384, 96, 32), ("f12e", 96, 576, 32), ("f12p", 576, 96, 32),
      ("f14p", 576, 160, 16), ("f15e", 160, 960, 16), ("f15p", 960, 160, 16), ("f17p", 960, 320, 16), ("f18", 320, 1280, 16)]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, cin, cout, H in SH:
    for bwd in (False, True):
        ci, co = (cout, cin) if bwd else (cin, cout)         # dgrad: K = Cout, M = Cin
        x = torch.randn(N, ci, H, H, device=DEV); y = torch.randn(N, ci, H, H, device=DEV)
        w = torch.randn(cout, cin, 1, 1, device=DEV) * 0.05
        cst = torch.rand(ci, SC_CST, device=DEV)
        src = make_src(x, ci, SRC_BNBWD, act=ACT_RELU6, cst=cst, aux=y) if bwd else make_src(x, ci, SRC_AFFINE, act=ACT_RELU6, cst=cst)
        co_t = _pick_cot(co, 1)
        wp = pack(w, co_t, 1 if bwd else 0)
        outs = [torch.empty(N, co, H, H, device=DEV)]
        t0 = timeit(lambda: conv_mfma([src], wp, N, H, H, co, 1, co_t, want_stats=not bwd, outs=outs))
        a = outs[0].clone()
        t1 = timeit(lambda: conv_mfma([src], wp, N, H, H, co, 1, co_t, want_stats=not bwd, outs=outs, ksplit=True))
        err = float((outs[0] - a).abs().max() / a.abs().max())
        print(f"{name:5s} {'dgrad' if bwd else 'fwd  '} K={ci:5d} M={co:5d} {H:3d}^2 co_t={co_t}  tile128 {t0:7.1f} us | ksplit {t1:7.1f} us | x{t0/t1:4.2f} diff {err:.1e}")
