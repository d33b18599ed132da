"""does a latency-bound low-resolution launch finish sooner as two half-batch launches on two streams?  python tools/bench_split_streams.py"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import ctypes as C
import torch
from starcop_amd import _lib
from starcop_amd._lib import SC_CST, SRC_AFFINE, ACT_RELU6, STAT_PW3, check, make_src, ptr, sc_conv_args
from hip_ops import pack_pw3
lib = _lib.load()
dev = "cuda"


def args(x, cst, wpk, out, stats, N, H, W, cin, cout):
    a = sc_conv_args()
    a.nsrc = 1
    a.src[0] = make_src(x, cin, SRC_AFFINE, act=ACT_RELU6, cst=cst)
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, cout, 1, 32
    a.out0, a.out1, a.csplit = out.data_ptr(), None, cout
    a.stats = stats.data_ptr()
    return a


for cin, cout, hw in ((64, 384, 32), (384, 64, 32), (160, 960, 16), (960, 160, 16), (96, 576, 32)):
    N = 16
    x = torch.randn(N, cin, hw, hw, device=dev); w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    cst = torch.rand(cin, SC_CST, device=dev) + 0.5
    wpk = pack_pw3(w, 0)
    out = torch.empty(N, cout, hw, hw, device=dev)
    rows = lib.sc_stat_rows(STAT_PW3, N, hw, hw)
    stats = torch.empty(rows, cout, 2, device=dev)
    full = args(x, cst, wpk, out, stats, N, hw, hw, cin, cout)
    h0 = args(x[:N // 2], cst, wpk, out[:N // 2], stats[:rows // 2], N // 2, hw, hw, cin, cout)
    h1 = args(x[N // 2:], cst, wpk, out[N // 2:], stats[rows // 2:], N // 2, hw, hw, cin, cout)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def run_full():
        check(lib.sc_conv1x1_pw3(C.byref(full), C.c_void_p(main.cuda_stream)))

    def run_split():
        s2.wait_stream(main)
        check(lib.sc_conv1x1_pw3(C.byref(h0), C.c_void_p(main.cuda_stream)))
        check(lib.sc_conv1x1_pw3(C.byref(h1), C.c_void_p(s2.cuda_stream)))
        main.wait_stream(s2)

    def chain(fn, reps=200):          # a dependent chain of launches, as in the forward pass
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    print(f"{cin:4d} -> {cout:4d} @ {hw}^2: one launch {chain(run_full):6.1f} us, two half-batch launches on two streams {chain(run_split):6.1f} us")
