"""head forward (sc_head_conv_fwd, Cin = 16) at the bench shape: the 16-byte staging kernel (k_head_fwd16v) against the 4-byte one
(k_head_fwd16), which the entry point falls back to for a source that is not 16-byte aligned -- so the same library times both.
usage: python tools/bench_head.py [N H W]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from starcop_amd import _lib
from starcop_amd._lib import ACT_RELU, SRC_AFFINE, check, make_src, ptr, stream
from hip_ops import cst_affine

N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 512, 512)
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
buf = torch.randn(N * 16 * H * W + 4, generator=g).to(dev)
w, b = (torch.randn(1, 16, 3, 3, generator=g) * 0.3).to(dev), torch.tensor([0.37], device=dev)
cst = cst_affine(torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.2)
out = torch.empty(N, 1, H, W, device=dev)
res = {}
for name, off in (("16-byte staging", 0), ("4-byte staging", 1), ("16-byte staging (again)", 0)):
    x = buf[off:off + N * 16 * H * W].view(N, 16, H, W)
    if off:
        x.copy_(buf[:N * 16 * H * W].view(N, 16, H, W).clone())
    src = make_src(x, 16, SRC_AFFINE, act=ACT_RELU, cst=cst)
    for _ in range(5):
        check(lib.sc_head_conv_fwd(C.byref(src), ptr(w), ptr(b), ptr(out), N, 16, H, W, stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        check(lib.sc_head_conv_fwd(C.byref(src), ptr(w), ptr(b), ptr(out), N, 16, H, W, stream()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    res[name] = out.clone()
    print(f"{name:26s} {us:8.1f} us   {N * 16 * H * W * 4 / us / 1e6:6.2f} TB/s of the source")
    if off:
        buf[:N * 16 * H * W].copy_(x.reshape(-1).clone())
print("max |difference| between the two kernels:", float((res["16-byte staging"] - res["4-byte staging"]).abs().max()))
