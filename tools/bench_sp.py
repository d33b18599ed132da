"""sub-pixel decoder conv1 (sc_conv3x3_sp [+ skip part on sc_conv3x3_bx3]) against the 3x3 form (sc_conv3x3_bx3 with an up-sampled
source): python tools/bench_sp.py [batch]  -- results and time per launch on the decoder's conv1 shapes"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack_bx3
from starcop_amd import _lib
from starcop_amd._lib import SRC_AFFINE, ACT_RELU, ACT_NONE, SC_CST, TERMS_F16X2, make_src, sc_conv_args, check, ptr, stream

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def sp_conv(srcs, wsp, N, H, W, cout, out, stats):
    a = sc_conv_args()
    a.nsrc = len(srcs)
    for i, s_ in enumerate(srcs):
        a.src[i] = s_
    a.wpk = wsp.data_ptr(); a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, cout, 3, 32
    a.out0 = out.data_ptr(); a.out1 = None; a.csplit = cout; a.accum0 = 0; a.accum1 = 0
    a.add0 = None; a.add1 = None; a.stats = stats.data_ptr() if stats is not None else None
    a.terms = TERMS_F16X2; a.down0 = 0; a.absmax = None
    check(lib.sc_conv3x3_sp(C.byref(a), stream()))


# (name, up channels, skip channels, cout, H out)
SHAPES = [("d0a", 1280, 96, 256, 32), ("d1a", 256, 32, 128, 64), ("d2a", 128, 24, 64, 128), ("d3a", 64, 16, 32, 256), ("d4a", 32, 0, 16, 512)]
for name, cu, cs, cout, H in ([] if os.environ.get("SP_WGRAD_ONLY") else SHAPES):
    W = H
    xl = torch.randn(N, cu, H // 2, W // 2, device=DEV)
    xs = torch.randn(N, max(cs, 1), H, W, device=DEV)
    w = torch.randn(cout, cu + cs, 3, 3, device=DEV) * 0.05
    cl = torch.rand(cu, SC_CST, device=DEV); cl[:, 1] -= 0.5
    csk = torch.rand(max(cs, 1), SC_CST, device=DEV)
    s_up = make_src(xl, cu, SRC_AFFINE, act=ACT_RELU, up=1, cst=cl)
    s_sk = make_src(xs, cs, SRC_AFFINE, act=ACT_NONE, cst=csk)
    co_t = 64 if cout > 32 else 32
    wb = pack_bx3(w, co_t, 0, TERMS_F16X2)
    ref = [torch.empty(N, cout, H, W, device=DEV)]
    srcs = [s_up, s_sk] if cs else [s_up]
    _, st_ref = conv_mfma(srcs, wb, N, H, W, cout, 3, co_t, want_stats=True, outs=ref, bx3=True, terms=TERMS_F16X2)
    t_ref = 0.0 if os.environ.get("SP_ONLY") else timeit(lambda: conv_mfma(srcs, wb, N, H, W, cout, 3, co_t, want_stats=True, outs=ref, bx3=True, terms=TERMS_F16X2))
    # sub-pixel: one launch, the skip channels as parity planes
    wsp = torch.empty(lib.sc_packed_weight_floats_sp(cout, cu, cs), device=DEV)
    check(lib.sc_pack_weights_sp(ptr(w), ptr(wsp), cout, cu, cs, TERMS_F16X2, stream()))
    out = torch.empty(N, cout, H, W, device=DEV)
    rows = lib.sc_sp_stat_rows(N, H, W, cout)
    stats = torch.full((rows, cout, 2), float("nan"), device=DEV)
    run = lambda: sp_conv(srcs, wsp, N, H, W, cout, out, stats)
    run()
    torch.cuda.synchronize()
    err = float((out - ref[0]).abs().max() / ref[0].abs().max())
    s0, s1 = stats.double().sum(0), st_ref.double().sum(0)
    serr = float(((s0 - s1).abs() / s1.abs().clamp_min(1e-3)).max())
    t_sp = timeit(run)
    t_up = t_sp
    if cs and not os.environ.get("SP_ONLY"):
        wsp_up = torch.zeros(lib.sc_packed_weight_floats_sp(cout, cu, 0), device=DEV)
        t_up = timeit(lambda: sp_conv(srcs[:1], wsp_up, N, H, W, cout, out, stats))
    if os.environ.get("SP_ONLY"):
        print(f"{name}: sub-pixel {t_sp:7.1f} us", flush=True)
        continue
    flop = 2.0 * N * H * W * (cu + cs) * cout * 9
    print(f"{name} {cu}+{cs}->{cout} {H}^2: 3x3 {t_ref:7.1f} us ({flop/t_ref/1e6:6.1f} TF) | sub-pixel {t_sp:7.1f} us ({flop/t_sp/1e6:6.1f} TF alg.; up channels alone {t_up:7.1f}) "
          f"x{t_ref/t_sp:4.2f} | out diff {err:.1e} stats diff {serr:.1e}", flush=True)

# ---- data gradient w.r.t. the up-sampled source: sub-pixel (sc_conv3x3_sp_dgrad) against sc_conv3x3_bx3(down0) on those channels
if not os.environ.get("SP_ONLY") and not os.environ.get("SP_WGRAD_ONLY"):
    from hip_ops import conv_sp_dgrad, pack_spd
    from starcop_amd._lib import SRC_BNBWD
    print("--- data gradient of the up-sampled channels ---")
    for name, cu, cs, cout, H in SHAPES:
        W = H
        g, yr = torch.randn(N, cout, H, W, device=DEV), torch.randn(N, cout, H, W, device=DEV)
        w = torch.randn(cout, cu + cs, 3, 3, device=DEV) * 0.05
        cst = torch.rand(cout, SC_CST, device=DEV)
        amax = (cst[:, 2][None, :, None, None] * g).abs().max().reshape(1)
        src = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=yr)
        co_t = 64 if cu % 64 == 0 else 32
        wu = pack_bx3(w[:, :cu].contiguous(), co_t, 1, TERMS_F16X2)
        ref = [torch.empty(N, cu, H // 2, W // 2, device=DEV)]
        f3 = lambda: conv_mfma([src], wu, N, H, W, cu, 3, co_t, outs=ref, bx3=True, terms=TERMS_F16X2, down0=True, absmax=amax)
        t_ref = timeit(f3)
        wsd = pack_spd(w, cu)
        out = torch.zeros(N, cu, H // 2, W // 2, device=DEV)
        fs = lambda: conv_sp_dgrad(src, wsd, N, H, W, cu, absmax=amax, accum_into=None)
        o = fs(); torch.cuda.synchronize()
        err = float((o - ref[0]).abs().max() / ref[0].abs().max())
        t_sp = timeit(fs)
        extra = ""
        if cs:
            # the skip channels' gradient: its own 3x3 launch on a 32-wide pack (what the network runs beside either form) ...
            wsk = pack_bx3(w[:, cu:].contiguous(), 32, 1, TERMS_F16X2)
            osk = [torch.empty(N, cs, H, W, device=DEV)]
            t_sk = timeit(lambda: conv_mfma([src], wsk, N, H, W, cs, 3, 32, outs=osk, bx3=True, terms=TERMS_F16X2, absmax=amax))
            extra = f" | skip channels' 3x3 launch {t_sk:6.1f} us"
            if lib.sc_spd_vskip_ok(cu, cs):      # ... or riding along as virtual channels of the sub-pixel launch
                wv = pack_spd(w, cu, vskip=True)
                fv = lambda: conv_sp_dgrad(src, wv, N, H, W, cu, absmax=amax, cskip=cs)
                ou, ok_ = fv(); torch.cuda.synchronize()
                e2 = max(float((ou - ref[0]).abs().max() / ref[0].abs().max()), float((ok_ - osk[0]).abs().max() / osk[0].abs().max()))
                t_v = timeit(fv)
                extra += f" | ONE launch for both (virtual skip channels) {t_v:6.1f} us vs {t_ref + t_sk:6.1f} (diff {e2:.1e})"
            else:                                # ... or as additional channel tiles of it (skip tiles)
                wt = pack_spd(w, cu, skip_tiles=True)
                ft = lambda: conv_sp_dgrad(src, wt, N, H, W, cu, absmax=amax, cskip=cs)
                ou, ok_ = ft(); torch.cuda.synchronize()
                e2 = max(float((ou - ref[0]).abs().max() / ref[0].abs().max()), float((ok_ - osk[0]).abs().max() / osk[0].abs().max()))
                t_t = timeit(ft)
                extra += f" | ONE launch for both (skip tiles) {t_t:6.1f} us vs sub-pixel + skip launch {t_sp + t_sk:6.1f} (diff {e2:.1e})"
        print(f"{name}.dgrad {cout}->{cu} {H}^2: 3x3 + down-sum {t_ref:7.1f} us | sub-pixel {t_sp:7.1f} us x{t_ref/t_sp:4.2f} | diff {err:.1e}{extra}", flush=True)

# ---- weight gradient of the up-sampled channels: box-sum GEMM (sc_conv3x3_sp_wgrad) against sc_conv3x3_wgrad_bx3 with an up-sampled source
if not os.environ.get("SP_ONLY"):
    from hip_ops import wgrad_mfma, wgrad_sp
    from starcop_amd._lib import SRC_BNBWD
    print("--- weight gradient (up-sampled channels) ---")
    for name, cu, cs, cout, H in SHAPES:
        W = H
        g, yr = torch.randn(N, cout, H, W, device=DEV), torch.randn(N, cout, H, W, device=DEV)
        xl = torch.randn(N, cu, H // 2, W // 2, device=DEV)
        cst = torch.rand(cout, SC_CST, device=DEV)
        cl = torch.rand(cu, SC_CST, device=DEV); cl[:, 1] -= 0.5
        amax = (cst[:, 2][None, :, None, None] * g).abs().max().reshape(1)
        dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=yr)
        s_up = make_src(xl, cu, SRC_AFFINE, act=ACT_RELU, up=1, cst=cl)
        use_bx3 = cout >= 32 and cu >= 32
        f3 = lambda: wgrad_mfma(dys, [s_up], N, H, W, cout, cu, 3, bx3=use_bx3, terms=TERMS_F16X2 if use_bx3 else 0, absmax=amax)
        ref = f3(); t_ref = timeit(f3)
        dw = torch.empty(cout, cu, 3, 3, device=DEV)
        fs = lambda: wgrad_sp(dys, s_up, N, H, W, cout, cu, absmax=amax, dw=dw)
        fs(); torch.cuda.synchronize()
        err = float((dw - ref).abs().max() / ref.abs().max())
        t_sp = timeit(fs)
        extra = ""
        if cs:
            xs = torch.randn(N, cs, H, W, device=DEV)
            s_sk = make_src(xs, cs, SRC_AFFINE, act=ACT_NONE, cst=torch.rand(cs, SC_CST, device=DEV))
            bx_all = cout >= 32 and cu + cs >= 32
            t_all = timeit(lambda: wgrad_mfma(dys, [s_up, s_sk], N, H, W, cout, cu + cs, 3, bx3=bx_all, terms=TERMS_F16X2 if bx_all else 0, absmax=amax))
            bx_sk = cout >= 32 and cs >= 32
            t_sk = timeit(lambda: wgrad_mfma(dys, [s_sk], N, H, W, cout, cs, 3, bx3=bx_sk, terms=TERMS_F16X2 if bx_sk else 0, absmax=amax))
            extra = f" | in-network: ONE 3x3 launch over all {cu + cs} channels {t_all:6.1f} us; the {cs} skip channels alone {t_sk:6.1f} us"
        print(f"{name}.wgrad {cout}x{cu} {H}^2: 3x3 form {t_ref:7.1f} us | box-sum GEMM {t_sp:7.1f} us x{t_ref/t_sp:4.2f} | diff {err:.1e}{extra}", flush=True)
