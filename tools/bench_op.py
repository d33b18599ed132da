"""Micro-bench of single conv ops through the C ABI (for rocprofv3 --pmc runs): python tools/bench_op.py <op> ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack, wgrad_mfma
from starcop_amd._lib import SRC_RAW, SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src

op, N, cin, cout, H, W, ks = sys.argv[1], *map(int, sys.argv[2:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
torch.manual_seed(0)
x = torch.randn(N, cin, H, W, device=DEV)
w = torch.randn(cout, cin, ks, ks, device=DEV) * 0.1
g = torch.randn(N, cout, H, W, device=DEV)
y = torch.randn(N, cout, H, W, device=DEV)
cst = torch.rand(cout, SC_CST, device=DEV)
cin_cst = torch.rand(cin, SC_CST, device=DEV)
flop = 2.0 * N * H * W * cin * cout * ks * ks


def run():
    if op == "headwgrad":
        import ctypes as C
        from starcop_amd import _lib
        from starcop_amd._lib import check, ptr, stream
        lib = _lib.load()
        nws = lib.sc_head_wgrad_workspace_floats(N, cin, H, W)
        ws = torch.empty(nws, device=DEV); dw = torch.empty(1, cin, 3, 3, device=DEV); db = torch.empty(1, device=DEV)
        src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cin_cst)
        check(lib.sc_head_conv_wgrad(ptr(g), C.byref(src), ptr(ws), nws, ptr(dw), ptr(db), N, cin, H, W, stream()))
        return dw
    if op == "wgrad":
        dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y)
        return wgrad_mfma(dys, [make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cin_cst)], N, H, W, cout, cin, ks)
    co_t = 64 if cout > 32 else 32
    wpk = pack(w, co_t, 0)
    return conv_mfma([make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cin_cst)], wpk, N, H, W, cout, ks, co_t, want_stats=(op == "fwdstats"))


run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{op} N={N} {cin}->{cout} {H}x{W} k{ks}: {dt*1e3:.3f} ms  {flop/dt/1e12:.1f} TFLOP/s")
