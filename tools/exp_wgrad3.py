"""k_wgrad3_bx3 alone on the decoder's shapes at batch 16 (two-fp16-term mode, BatchNorm-backward gradient source: the pipelined kernel),
for timing and for rocprofv3 --pmc passes of elimination builds (STARCOP_HIP_LIB=...):  python tools/exp_wgrad3.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, wgrad_mfma
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 16
torch.manual_seed(0)
for name, cin, cout, H in [("d1b", 128, 128, 64), ("d2a", 152, 64, 128), ("d2b", 64, 64, 128), ("d3a", 80, 32, 256), ("d3b", 32, 32, 256)]:
    x = torch.randn(N, cin, H, H, device=DEV)
    g = torch.randn(N, cout, H, H, device=DEV) * 1e-3
    y = torch.randn(N, cout, H, H, device=DEV)
    cst = torch.rand(cout, SC_CST, device=DEV)
    cstx = torch.rand(cin, SC_CST, device=DEV)
    amax = torch.tensor([8e-3], device=DEV)
    dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y)
    src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cstx)
    fn = lambda: wgrad_mfma(dys, [src], N, H, H, cout, cin, 3, bx3=True, terms=4, absmax=amax)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / REPS
    print(f"{name:5s} {cin:4d}->{cout:4d} {H:3d}^2  {t*1e3:7.1f} us (incl. reduce)  {2.0*N*H*H*cin*cout*9/t/1e9:6.1f} TF", flush=True)
