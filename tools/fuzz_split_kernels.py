"""random-shape sweep of the split 3x3 kernels against the fp32-MFMA kernels of the same ABI: python tools/fuzz_split_kernels.py [cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import dev, DEV, conv_mfma, pack, pack_bx3, relerr, wgrad_mfma
from starcop_amd import _lib
from starcop_amd._lib import *
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
bad = 0; checks = 0
for it in range(cases):
    N = rng.choice([1, 2, 3]); H = rng.choice([2, 4, 6, 8, 10, 16, 18, 34]); W = rng.choice([2, 4, 12, 32, 34, 62, 64, 70])
    cin = rng.choice([16, 32, 48, 80, 160, 272]); cout = rng.choice([8, 16, 32, 64, 96, 136]); co_t = 64 if cout > 32 else 32
    bnb = rng.random() < 0.5
    x = rnd(N, cin, H, W, seed=it * 7 + 1); w = rnd(cout, cin, 3, 3, seed=it * 7 + 2, scale=0.2)
    cst = torch.rand(cin, SC_CST, generator=torch.Generator().manual_seed(it * 7 + 3)) + 0.5
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, cst=dev(cst))
    srcs = [src]
    two = cin >= 48 and H % 2 == 0 and W % 2 == 0 and rng.random() < 0.5          # concat of an upsampled source and a raw skip
    if two:
        c0 = (cin - 16) // 16 * 16
        prev, skip = rnd(N, c0, H // 2, W // 2, seed=it * 7 + 6), rnd(N, cin - c0, H, W, seed=it * 7 + 7)
        srcs = [make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=dev(cst[:c0].contiguous())), make_src(dev(skip), cin - c0, SRC_RAW)]
    tag = f"case {it}: N{N} {cin}->{cout} {H}x{W} bnb={bnb} two={two}"
    try:
        if cout >= 32:
            (o1,), s1 = conv_mfma(srcs, pack_bx3(dev(w), co_t, 0, TERMS_F16X2), N, H, W, cout, 3, co_t, bx3=True, terms=TERMS_F16X2, want_stats=True)
            (o0,), s0 = conv_mfma(srcs, pack(dev(w), co_t, 0), N, H, W, cout, 3, co_t, want_stats=True)
            e = relerr(o1, o0.double().cpu()); es = float((s1.sum(0) - s0.sum(0)).abs().max() / s0.sum(0).abs().max())
            checks += 1
            if not (e < 5e-6 and es < 1e-4): bad += 1; print("FWD", tag, e, es)
        # gradient side
        g, y = rnd(N, cout, H, W, seed=it * 7 + 4, scale=1e-3), rnd(N, cout, H, W, seed=it * 7 + 5)
        cb = torch.zeros(cout, SC_CST); cb[:, 0] = 1.0; cb[:, 2] = 1.0 + 0.1 * torch.arange(cout) / cout; cb[:, 3] = 1e-5; cb[:, 4] = 1e-5
        amax = torch.tensor([2e-3 * 6], device=DEV)
        dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cb), aux=dev(y)) if bnb else make_src(dev(g), cout, SRC_RAW)
        if cin >= 32 and cout >= 32:
            cbt = 64 if cin > 32 else 32
            (d1,), _ = conv_mfma([dsrc], pack_bx3(dev(w), cbt, 1, TERMS_F16X2), N, H, W, cin, 3, cbt, bx3=True, terms=TERMS_F16X2, absmax=amax)
            (d0,), _ = conv_mfma([dsrc], pack(dev(w), cbt, 1), N, H, W, cin, 3, cbt)
            e = relerr(d1, d0.double().cpu())
            checks += 1
            if not e < 2e-5: bad += 1; print("DGRAD", tag, e)
            w1 = wgrad_mfma(dsrc, srcs, N, H, W, cout, cin, 3, bx3=True, terms=TERMS_F16X2, absmax=amax)
            w0 = wgrad_mfma(dsrc, srcs, N, H, W, cout, cin, 3)
            e = relerr(w1, w0.double().cpu())
            checks += 1
            if not e < 2e-5: bad += 1; print("WGRAD", tag, e)
        if cout <= 16 and cin in (16, 32) and W % 2 == 0:
            a = sc_wgrad_args(); a.dy = dsrc; a.nsrc = 1; a.src[0] = src
            a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, cout, cin, 3
            n = max(lib.sc_wgrad_thin16_workspace_floats(N, H, W, cout, cin), lib.sc_wgrad_workspace_floats(N, H, W, cout, cin, 3))
            ws = torch.empty(n, device=DEV); dw = torch.empty(cout, cin, 3, 3, device=DEV)
            a.part, a.part_floats, a.dw, a.terms, a.absmax = ws.data_ptr(), n, dw.data_ptr(), TERMS_F16X2, amax.data_ptr()
            check(lib.sc_conv3x3_wgrad_thin16(C.byref(a), stream())); t1 = dw.clone()
            check(lib.sc_conv2d_wgrad_mfma(C.byref(a), stream()))
            e = float((t1 - dw).abs().max() / dw.abs().max())
            checks += 1
            if not e < 2e-5: bad += 1; print("THINW", tag, e)
    except Exception as ex:
        bad += 1; print("EXC", tag, type(ex).__name__, str(ex)[:160])
print(f"{bad} problems in {checks} checks of {cases} cases")
