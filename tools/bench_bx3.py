"""fp32-MFMA vs split-bf16 3x3 conv on the decoder shapes (B=16, 512x512 tiles): ms and TFLOP/s per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack, pack_bx3
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TERMS = int(os.environ.get("SC_TERMS", "4"))     # 4 = two fp16 terms (the default mode), 3 / 2 / 1 = bf16 terms
# (name, cin, cout, H)
SHAPES = [("d0a", 1376, 256, 32), ("d0b", 256, 256, 32), ("d1a", 288, 128, 64), ("d1b", 128, 128, 64), ("d2a", 152, 64, 128),
          ("d2b", 64, 64, 128), ("d3a", 80, 32, 256), ("d3b", 32, 32, 256), ("d0a.dgrad", 256, 1376, 32), ("d1a.dgrad", 128, 288, 64),
          ("d2a.dgrad", 64, 152, 128), ("d3a.dgrad", 32, 80, 256), ("d4a.dgrad", 16, 32, 512)]
torch.manual_seed(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, cin, cout, H in SHAPES:
    W = H
    x = torch.randn(N, cin, H, W, device=DEV)
    y = torch.randn(N, cin, H, W, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    cst = torch.rand(cin, SC_CST, device=DEV)
    bwd = name.endswith("dgrad")
    src = make_src(x, cin, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y) if bwd else make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cst)
    flop = 2.0 * N * H * W * cin * cout * 9
    co_t = int(os.environ.get("SC_COT", "0")) or (64 if cout > 32 else 32)
    wf, wb = pack(w, co_t, 0), pack_bx3(w, co_t, 0, TERMS)
    outs = [torch.empty(N, cout, H, W, device=DEV)]
    t32 = timeit(lambda: conv_mfma([src], wf, N, H, W, cout, 3, co_t, want_stats=not bwd, outs=outs))
    o32 = outs[0].clone()
    tbx = timeit(lambda: conv_mfma([src], wb, N, H, W, cout, 3, co_t, want_stats=not bwd, outs=outs, bx3=True, terms=TERMS))
    err = float((outs[0] - o32).abs().max() / o32.abs().max())
    print(f"{name:10s} {cin:5d}->{cout:4d} {H:3d}^2  f32 {t32:7.3f} ms {flop/t32/1e9:6.1f} TF | bx3 {tbx:7.3f} ms {flop/tbx/1e9:6.1f} TF | x{t32/tbx:4.2f}  diff {err:.1e}")

from hip_ops import wgrad_mfma
print("--- weight gradient ---")
for name, cin, cout, H in [("d0a", 1376, 256, 32), ("d0b", 256, 256, 32), ("d1a", 288, 128, 64), ("d1b", 128, 128, 64), ("d2a", 152, 64, 128),
                           ("d2b", 64, 64, 128), ("d3a", 80, 32, 256), ("d3b", 32, 32, 256)]:
    W = H
    x = torch.randn(N, cin, H, W, device=DEV)
    g = torch.randn(N, cout, H, W, device=DEV)
    y = torch.randn(N, cout, H, W, device=DEV)
    cst = torch.rand(cout, SC_CST, device=DEV)
    cstx = torch.rand(cin, SC_CST, device=DEV)
    dys = make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y)
    src = make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cstx)
    flop = 2.0 * N * H * W * cin * cout * 9
    t32 = timeit(lambda: wgrad_mfma(dys, [src], N, H, W, cout, cin, 3))
    tbx = timeit(lambda: wgrad_mfma(dys, [src], N, H, W, cout, cin, 3, bx3=True, terms=TERMS))
    a, b = wgrad_mfma(dys, [src], N, H, W, cout, cin, 3), wgrad_mfma(dys, [src], N, H, W, cout, cin, 3, bx3=True, terms=TERMS)
    err = float((a - b).abs().max() / a.abs().max())
    print(f"{name:10s} {cin:5d}->{cout:4d} {H:3d}^2  f32 {t32:7.3f} ms {flop/t32/1e9:6.1f} TF | bx3 {tbx:7.3f} ms {flop/tbx/1e9:6.1f} TF | x{t32/tbx:4.2f}  diff {err:.1e}")
