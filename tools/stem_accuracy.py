import sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from starcop_amd import _lib
from starcop_amd._lib import SRC_RAW, SRC_NORM, SC_CST, STAT_STEM, check, make_src, ptr, stream
from hip_ops import DEV, dev
lib = _lib.load()
torch.manual_seed(0)
N, Cin, H, W = 4, 4, 128, 128
x = torch.rand(N, Cin, H, W) * 2.0
w = torch.randn(32, Cin, 3, 3) * 0.3
ref = F.conv2d(x.double(), w.double(), stride=2, padding=1)
Ho = Wo = 64
obuf = torch.zeros(N * 32 * Ho * Wo + 4, device=DEV)
for off, name in ((0, "MFMA 16x16x4"), (1, "VALU fma chain")):
    out = obuf[off:off + N * 32 * Ho * Wo].view(N, 32, Ho, Wo)
    check(lib.sc_stem_conv_fwd(C.byref(make_src(dev(x), Cin, SRC_RAW)), ptr(dev(w)), ptr(out), N, Cin, H, W, None, stream()))
    d = (out.double().cpu() - ref)
    print(f"{name:16s} max |err| {float(d.abs().max()):.3e}   rms err {float(d.pow(2).mean().sqrt()):.3e}   mean err {float(d.mean()):+.3e}   (max |ref| {float(ref.abs().max()):.2f})")
o32 = F.conv2d(x, w, stride=2, padding=1)
d = o32.double() - ref
print(f"{'torch fp32 CPU':16s} max |err| {float(d.abs().max()):.3e}   rms err {float(d.pow(2).mean().sqrt()):.3e}   mean err {float(d.mean()):+.3e}")
