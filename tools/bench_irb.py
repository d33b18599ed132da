"""sc_irb_eval alone on the encoder's stride-1 block shapes (batch 16 planes of a 512 x 512 tile, or one 1280 x 1248 scene): us per launch
python tools/bench_irb.py [reps]      (STARCOP_HIP_LIB=... for elimination builds, tools/build_exp_irb.sh)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, cst_affine, dev, pack_pw3
from starcop_amd import _lib
from starcop_amd._lib import SRC_RAW, check, make_src, sc_irb_args, stream
lib = _lib.load(); REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [("f7 s2", 16, 32, 192, 64, 64, 64), ("f14 s2", 16, 96, 576, 160, 32, 32), ("f5", 16, 32, 192, 32, 64, 64), ("f8", 16, 64, 384, 64, 32, 32), ("f11", 16, 64, 384, 96, 32, 32), ("f12", 16, 96, 576, 96, 32, 32),
          ("f15", 16, 160, 960, 160, 16, 16), ("f17", 16, 160, 960, 320, 16, 16),
          ("scene f8", 1, 64, 384, 64, 80, 78), ("scene f12", 1, 96, 576, 96, 80, 78), ("scene f15", 1, 160, 960, 160, 40, 39)]
g = torch.Generator().manual_seed(0)
for name, N, Cin, hid, Cout, H, W in SHAPES:
    stride = 2 if name.endswith("s2") else 1
    if not lib.sc_irb_supported(Cin, hid, Cout, H, W, stride):
        print(f"{name}: unsupported"); continue
    x = dev(torch.randn(N, Cin, H, W, generator=g))
    We, Wd, Wp = torch.randn(hid, Cin, generator=g) * 0.1, torch.randn(hid, 3, 3, generator=g) * 0.3, torch.randn(Cout, hid, generator=g) * 0.05
    a = sc_irb_args()
    a.x = make_src(x, Cin, SRC_RAW)
    wpe, wpp, wd = pack_pw3(dev(We[:, :, None, None]), 0), pack_pw3(dev(Wp[:, :, None, None]), 0), dev(Wd)
    ce, cd, cp = (cst_affine(torch.ones(c_), torch.zeros(c_)) for c_ in (hid, hid, Cout))
    out = torch.empty(N, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1, device=DEV); zmax = torch.zeros(1, device=DEV)
    res = int(Cin == Cout and stride == 1)
    a.wpk_expand, a.cst_expand, a.w_dw, a.cst_dw, a.wpk_project = wpe.data_ptr(), ce.data_ptr(), wd.data_ptr(), cd.data_ptr(), wpp.data_ptr()
    a.cst_project, a.out, a.z_absmax = (cp.data_ptr() if res else None), out.data_ptr(), (zmax.data_ptr() if res else None)
    a.N, a.Cin, a.hidden, a.Cout, a.H, a.W, a.stride, a.residual = N, Cin, hid, Cout, H, W, stride, res
    fn = lambda: check(lib.sc_irb_eval(C.byref(a), stream()))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / REPS * 1e3
    mac = N * H * W * (Cin * hid + 9 * hid + hid * Cout)
    print(f"{name:10s} {N:2d} x {Cin:3d}->{hid:3d}->{Cout:3d} @ {H}x{W}: {t:7.1f} us   {2 * mac / t / 1e6:6.1f} TF/s fp32-equivalent   {t / (hid // (64 if H * W <= 256 and hid % 64 == 0 else 32)):.2f} us/chunk", flush=True)
