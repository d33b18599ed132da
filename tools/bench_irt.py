"""time the four sweeps of the fused inverted-residual training execution (conv_irt.hip) at the network's shapes, batch 16:
python tools/bench_irt.py [--batch 16]"""
import sys, os, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import SC_CST, SRC_AFFINE, SRC_BNBWD, ACT_NONE, ACT_RELU6, check, make_src, ptr, sc_irt_args, stream
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=10)
a_ = ap.parse_args()
lib = _lib.load(); _lib.require_device()
dev = "cuda"
# name, Cin, hidden, H (= W) of the block input, stride
BLOCKS = [("features.2", 16, 96, 256, 2), ("features.4", 24, 144, 128, 2), ("features.7", 32, 192, 64, 2)]
N = a_.batch
def timeit(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = {}
for name, Cin, Hd, H, S in BLOCKS:
    W = H; Ho = H // S; Wo = W // S
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, Cin, H, W, device=dev, generator=g)
    We = torch.randn(Hd, Cin, device=dev, generator=g) * 0.3
    Wd = torch.randn(Hd, 3, 3, device=dev, generator=g) * 0.4
    cx = torch.zeros(Cin, SC_CST, device=dev); cx[:, 0] = 1.0
    ce = torch.zeros(Hd, SC_CST, device=dev); ce[:, 0] = 0.5; ce[:, 1] = 1.0; ce[:, 3] = 0.5
    cb = torch.zeros(Hd, SC_CST, device=dev); cb[:, 0] = 0.5; cb[:, 1] = 1.0; cb[:, 2] = 0.5; cb[:, 3] = 0.01
    cdb = cb.clone()
    a = sc_irt_args(); a.x = make_src(x, Cin, SRC_AFFINE, act=ACT_NONE, cst=cx)
    a.w_expand, a.w_dw, a.cst_expand = We.data_ptr(), Wd.data_ptr(), ce.data_ptr()
    a.N, a.Cin, a.hidden, a.H, a.W, a.stride = N, Cin, Hd, H, W, S
    r0, r1, r2 = lib.sc_irt_rows(0, N, H, W, S), lib.sc_irt_rows(1, N, H, W, S), lib.sc_irt_bwd_rows(N, Hd, H, W)
    st0 = torch.empty(r0, Hd, 2, device=dev); st1 = torch.empty(r1, Hd, 2, device=dev)
    d = torch.empty(N, Hd, Ho, Wo, device=dev); gd = torch.randn(N, Hd, Ho, Wo, device=dev, generator=g)
    es = torch.empty(r2, Hd, 2, dtype=torch.float64, device=dev); dwa = torch.zeros(Hd, 9, dtype=torch.float64, device=dev)
    work = torch.empty(lib.sc_irt_bwd_workspace_floats(N, Cin, Hd, H, W), device=dev)
    dx = torch.empty(N, Cin, H, W, device=dev); dWe = torch.empty(Hd, Cin, device=dev)
    st = stream()
    check(lib.sc_irt_fwd(C.byref(a), ptr(d), ptr(st1), st))
    dy = make_src(gd, Hd, SRC_BNBWD, act=ACT_RELU6, cst=cdb, aux=d)
    t = [timeit(lambda: check(lib.sc_irt_expand_stats(C.byref(a), ptr(st0), st)), a_.reps),
         timeit(lambda: check(lib.sc_irt_fwd(C.byref(a), ptr(d), ptr(st1), st)), a_.reps),
         timeit(lambda: check(lib.sc_irt_bwd(C.byref(a), C.byref(dy), ptr(es), ptr(dwa), ptr(work), st)), a_.reps),
         timeit(lambda: check(lib.sc_irt_bwd_fix(C.byref(a), ptr(cb), ptr(work), ptr(dx), None, 0, st)), a_.reps),
         timeit(lambda: check(lib.sc_irt_xmoments(C.byref(a), ptr(work), st)), a_.reps),
         timeit(lambda: check(lib.sc_irt_wgrad_finalize(C.byref(a), ptr(cb), ptr(work), ptr(dWe), st)), a_.reps)]
    eb = N * Hd * H * W * 4 / 1e6
    print(f"{name:11s} Cin {Cin:3d} hid {Hd:3d} {H:3d}^2 s{S}  e = {eb:6.0f} MB | stats {t[0]:7.1f}  fwd {t[1]:7.1f}  bwd {t[2]:7.1f}  fix {t[3]:7.1f}  "
          f"xmom {t[4]:6.1f}  dwe {t[5]:6.1f} us | sum {sum(t):7.1f} us   rows {r0} {r1} {r2}")
