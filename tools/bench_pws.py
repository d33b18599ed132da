"""the large-plane pointwise forward launches of the network (sc_conv2d_mfma, ks = 1) at batch 16: the streaming kernel (k_pw_stream)
against the LDS-staged one (k_conv_mfma<1>, which the entry point falls back to for a source that is not 16-byte aligned), same library.
usage: python tools/bench_pws.py [stats=1]"""
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from starcop_amd._lib import ACT_NONE, ACT_RELU6, SRC_AFFINE, make_src
from hip_ops import DEV, conv_mfma, cst_affine, pack

want_stats = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
LAYERS = [("features.1 project", 32, 16, 256, ACT_RELU6), ("features.2 project", 96, 24, 128, ACT_RELU6), ("features.3 expand", 24, 144, 128, ACT_NONE),
          ("features.3 project", 144, 24, 128, ACT_RELU6), ("features.4 expand", 24, 144, 128, ACT_NONE)]
N = 16
g = torch.Generator().manual_seed(0)
tot = [0.0, 0.0]
for name, Cin, Cout, S, act in LAYERS:
    n = N * Cin * S * S
    buf = torch.randn(n + 4, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
    co_t = 32 if Cout <= 32 else 64
    wpk = pack(w, co_t, 0)
    cst = cst_affine(torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3)
    out = [torch.empty(N, Cout, S, S, device=DEV)]
    res = []
    for off in (0, 1):
        x = buf[off:off + n].view(N, Cin, S, S)
        src = make_src(x, Cin, SRC_AFFINE, act=act, cst=cst)
        for _ in range(3):
            conv_mfma([src], wpk, N, S, S, Cout, 1, co_t, want_stats=want_stats, outs=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            conv_mfma([src], wpk, N, S, S, Cout, 1, co_t, want_stats=want_stats, outs=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    mb = N * (Cin + Cout) * S * S * 4 / 1e6
    tot[0] += res[0]; tot[1] += res[1]
    print(f"{name:20s} {Cin:4d} -> {Cout:4d} @ {S}^2  {mb:6.0f} MB   streaming {res[0]:6.1f} us ({mb / res[0]:5.2f} TB/s)   LDS-staged {res[1]:6.1f} us ({mb / res[1]:5.2f} TB/s)")
print(f"sum: streaming {tot[0]:.1f} us, LDS-staged {tot[1]:.1f} us   (the timings include the helper's statistics allocation when stats=1)")
# ---- the projections' data gradients (BatchNorm-backward source: gradient + raw tensor read, few channels -> many)
from starcop_amd._lib import SC_CST, SRC_BNBWD
tot = [0.0, 0.0]
for name, Cin, Cout, S in [("features.1 project dgrad", 32, 16, 256), ("features.2 project dgrad", 96, 24, 128), ("features.3 project dgrad", 144, 24, 128)]:
    n = N * Cout * S * S
    gb, yb = torch.randn(n + 4, generator=g).to(DEV), torch.randn(n + 4, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(DEV)
    co_t = 32 if Cin <= 32 else 64
    wpk = pack(w, co_t, 1)
    cst = torch.zeros(Cout, SC_CST); cst[:, 0] = 1.0; cst[:, 2] = 1.0; cst[:, 3] = 0.01
    cst = cst.to(DEV)
    out = [torch.empty(N, Cin, S, S, device=DEV)]
    res = []
    for off in (0, 1):
        src = make_src(gb[off:off + n].view(N, Cout, S, S), Cout, SRC_BNBWD, act=ACT_NONE, cst=cst, aux=yb[off:off + n].view(N, Cout, S, S))
        for _ in range(3):
            conv_mfma([src], wpk, N, S, S, Cin, 1, co_t, outs=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            conv_mfma([src], wpk, N, S, S, Cin, 1, co_t, outs=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    mb = N * (2 * Cout + Cin) * S * S * 4 / 1e6
    tot[0] += res[0]; tot[1] += res[1]
    print(f"{name:26s} {Cout:4d} -> {Cin:4d} @ {S}^2  {mb:6.0f} MB   streaming {res[0]:6.1f} us ({mb / res[0]:5.2f} TB/s)   LDS-staged {res[1]:6.1f} us ({mb / res[1]:5.2f} TB/s)")
print(f"data gradients: streaming {tot[0]:.1f} us, LDS-staged {tot[1]:.1f} us")
