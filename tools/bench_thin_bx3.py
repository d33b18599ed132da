import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from hip_ops import DEV, conv_mfma, pack, pack_bx3, wgrad_mfma
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src
N = 16
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
HH = int(os.environ.get("THIN_H", "512"))
for name, cin, cout, H, up in [("d4a", 32, 16, HH, 1), ("d4b", 16, 16, HH, 0), ("d4b.dgrad", 16, 16, HH, 0)]:
    W = H
    x = torch.randn(N, cin, H >> up, W >> up, device=DEV)
    y = torch.randn(N, cin, H, W, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    cst = torch.rand(cin, SC_CST, device=DEV)
    bwd = name.endswith("dgrad")
    src = make_src(x, cin, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y) if bwd else make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cst, up=up)
    flop = 2.0 * N * H * W * cin * cout * 9
    outs = [torch.empty(N, cout, H, W, device=DEV)]
    t16 = timeit(lambda: conv_mfma([src], pack(w, 16, 0), N, H, W, cout, 3, 16, want_stats=not bwd, outs=outs))
    o16 = outs[0].clone()
    tbx = timeit(lambda: conv_mfma([src], pack_bx3(w, 32, 0, 4), N, H, W, cout, 3, 32, want_stats=not bwd, outs=outs, bx3=True, terms=4))
    err = float((outs[0] - o16).abs().max() / o16.abs().max())
    import ctypes as C
    from starcop_amd import _lib
    from starcop_amd._lib import sc_conv_args, check, ptr, stream, STAT_CONV3
    lib = _lib.load()
    wt = torch.empty(lib.sc_packed_weight_floats_thin16(cout, cin, 0), device=DEV)
    check(lib.sc_pack_weights_thin16(ptr(w), ptr(wt), cout, cin, 0, stream()))
    a = sc_conv_args(); a.nsrc = 1; a.src[0] = src; a.wpk = wt.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, cout, 3, 16
    a.out0 = outs[0].data_ptr(); a.csplit = cout; a.terms = 4
    st_ = torch.empty(lib.sc_stat_rows(STAT_CONV3, N, H, W), cout, 2, device=DEV)
    a.stats = None if bwd else st_.data_ptr()
    tth = timeit(lambda: check(lib.sc_conv3x3_thin16(C.byref(a), stream())))
    err2 = float((outs[0] - o16).abs().max() / o16.abs().max())
    print(f"{name:10s} {cin:3d}->{cout:3d} {H}^2  thin-fp32 {t16:.3f} ms | split(co_t=32) {tbx:.3f} ms  diff {err:.1e} | thin16-fp16x2 {tth:.3f} ms diff {err2:.1e}")
print("--- weight gradient ---")
for name, cin, cout, H, up in [("d4a", 32, 16, 512, 1), ("d4b", 16, 16, 512, 0), ("d3b", 32, 32, 256, 0)]:
    W = H
    x = torch.randn(N, cin, H >> up, W >> up, device=DEV)
    g = torch.randn(N, cout, H, W, device=DEV)
    y = torch.randn(N, cout, H, W, device=DEV)
    cst = torch.rand(cout, SC_CST, device=DEV)
    cstx = torch.rand(cin, SC_CST, device=DEV)
    dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y)
    src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cstx, up=up)
    t32 = timeit(lambda: wgrad_mfma(dys, [src], N, H, W, cout, cin, 3))
    try:
        tbx = timeit(lambda: wgrad_mfma(dys, [src], N, H, W, cout, cin, 3, bx3=True, terms=4))
        a, b = wgrad_mfma(dys, [src], N, H, W, cout, cin, 3), wgrad_mfma(dys, [src], N, H, W, cout, cin, 3, bx3=True, terms=4)
        err = float((a - b).abs().max() / a.abs().max())
    except Exception as e:
        tbx, err = float("nan"), str(e)[:80]
    print(f"{name:6s} {cin:3d}->{cout:3d} {H}^2  fp32 {t32:.3f} ms | split {tbx:.3f} ms  diff {err}")
