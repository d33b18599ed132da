"""thin 3x3 weight gradients (decoder.blocks.4 at 16 x 512^2): split-fp16 kernel vs the fp32-MFMA kernel"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import *
lib = _lib.load(); st = stream(); N, H = 16, 512
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, cin, up in (("d4a (32->16, upsampled input)", 32, 1), ("d4b (16->16)", 16, 0)):
    x = torch.randn(N, cin, H >> up, H >> up, device="cuda"); g = torch.randn(N, 16, H, H, device="cuda") * 1e-3; y = torch.randn(N, 16, H, H, device="cuda")
    cx = torch.zeros(cin, SC_CST, device="cuda"); cx[:, 0] = 1
    cd = torch.zeros(16, SC_CST, device="cuda"); cd[:, 0] = 1; cd[:, 2] = 1
    amax = torch.tensor([4e-3], device="cuda")
    a = sc_wgrad_args()
    a.dy = make_src(g, 16, SRC_BNBWD, act=ACT_RELU, cst=cd, aux=y); a.nsrc = 1
    a.src[0] = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, up=up, cst=cx)
    a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, H, 16, cin, 3
    n = max(lib.sc_wgrad_thin16_workspace_floats(N, H, H, 16, cin), lib.sc_wgrad_workspace_floats(N, H, H, 16, cin, 3))
    ws = torch.empty(n, device="cuda"); dw = torch.empty(16, cin, 3, 3, device="cuda")
    a.part, a.part_floats, a.dw = ws.data_ptr(), n, dw.data_ptr()
    a.terms, a.absmax = TERMS_F16X2, amax.data_ptr()
    t1 = timeit(lambda: check(lib.sc_conv3x3_wgrad_thin16(C.byref(a), st)))
    d1 = dw.clone()
    t0 = timeit(lambda: check(lib.sc_conv2d_wgrad_mfma(C.byref(a), st)))
    print(f"{name}: split-fp16 {t1*1e3:.0f} us, fp32 MFMA {t0*1e3:.0f} us; max rel diff {float((d1-dw).abs().max()/dw.abs().max()):.1e}")
