"""One conv shape through the C ABI a few times (for rocprofv3 --pmc): python tools/bench_one.py {fwd|dgrad|wgrad} cin cout H [bx3] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hip_ops import DEV, conv_mfma, pack, pack_bx3, wgrad_mfma
from starcop_amd._lib import SRC_AFFINE, SRC_BNBWD, ACT_RELU, SC_CST, make_src
op, cin, cout, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bx3 = len(sys.argv) > 5 and sys.argv[5] == "1"
N = int(sys.argv[6]) if len(sys.argv) > 6 else 16
W = H
torch.manual_seed(0)
x = torch.randn(N, cin, H, W, device=DEV); y = torch.randn(N, cin, H, W, device=DEV)
g = torch.randn(N, cout, H, W, device=DEV); yo = torch.randn(N, cout, H, W, device=DEV)
w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
cst = torch.rand(cin, SC_CST, device=DEV); csto = torch.rand(cout, SC_CST, device=DEV)
co_t = 64 if cout > 32 else 32
wp = pack_bx3(w, co_t, 0) if bx3 else pack(w, co_t, 0)
outs = [torch.empty(N, cout, H, W, device=DEV)]
for _ in range(3):
    if op == "wgrad":
        wgrad_mfma(make_src(g, cout, SRC_BNBWD, act=ACT_RELU, cst=csto, aux=yo), [make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cst)], N, H, W, cout, cin, 3, bx3=bx3)
    else:
        src = make_src(x, cin, SRC_BNBWD, act=ACT_RELU, cst=cst, aux=y) if op == "dgrad" else make_src(x, cin, SRC_AFFINE, act=ACT_RELU, cst=cst)
        conv_mfma([src], wp, N, H, W, cout, 3, co_t, want_stats=(op == "fwd"), outs=outs, bx3=bx3)
torch.cuda.synchronize()
