"""Thin 3x3 layers (decoder.blocks.4.*: 32->16 and 16->16 at 512^2, B=16) through k_conv_mfma16 / k_wgrad_mfma16 (for PMC runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack, wgrad_mfma
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src
N, H = 16, 512
for cin in (32, 16):
    x = torch.randn(N, cin, H, H, device=DEV); g = torch.randn(N, 16, H, H, device=DEV); y = torch.randn(N, 16, H, H, device=DEV)
    w = torch.randn(16, cin, 3, 3, device=DEV) * 0.1
    cst = torch.rand(cin, SC_CST, device=DEV); csto = torch.rand(16, SC_CST, device=DEV)
    src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cst)
    dys = make_src(g, 16, SRC_BNBWD, act=ACT_RELU, cst=csto, aux=y)
    wp = pack(w, 16, 0)
    outs = [torch.empty(N, 16, H, H, device=DEV)]
    def T(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
    t = T(lambda: conv_mfma([src], wp, N, H, H, 16, 3, 16, want_stats=True, outs=outs))
    fl = 2.0 * N * H * H * cin * 16 * 9
    by = 4.0 * N * H * H * (cin + 16)
    print(f"fwd {cin}->16: {t:.3f} ms  {fl/t/1e9:.1f} TF  {by/t/1e9:.2f} TB/s")
    t = T(lambda: wgrad_mfma(dys, [src], N, H, H, 16, cin, 3))
    print(f"wgrad {cin}->16: {t:.3f} ms  {fl/t/1e9:.1f} TF  {4.0*N*H*H*(cin+32)/t/1e9:.2f} TB/s")
