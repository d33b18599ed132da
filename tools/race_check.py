"""repeat-run determinism check of the 3x3 split kernels on one shape: python tools/race_check.py [cin cout H W reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_ops import *          # noqa: helpers (conv_mfma, pack_bx3, make_src, rnd, dev, ...)
from starcop_amd._lib import TERMS_F16X2
cin, cout, H, W, reps = (int(a) for a in (sys.argv[1:6] + [288, 128, 8, 12, 200][len(sys.argv) - 1:]))
from starcop_amd import _lib; _lib.load()
N = 2
x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
cst = torch.rand(cin, SC_CST) + 0.5
src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, cst=dev(cst))
wp = pack_bx3(dev(w), 64, 0, TERMS_F16X2)
ref = None; bad = 0
for i in range(reps):
    (out,), st = conv_mfma([src], wp, N, H, W, cout, 3, 64, bx3=True, terms=TERMS_F16X2, want_stats=True)
    torch.cuda.synchronize()
    if ref is None: ref, rst = out.clone(), st.clone()
    elif not (torch.equal(out, ref) and torch.equal(st, rst)):
        bad += 1
        d = (out - ref).abs()
        if bad <= 3: print("run", i, "differs: max", float(d.max()), "at", [int(v) for v in torch.nonzero(d == d.max())[0]], "count", int((d > 0).sum()), "stats equal", bool(torch.equal(st, rst)))
print(f"{bad} of {reps - 1} repeat runs differ from the first")
