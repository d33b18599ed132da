"""Known-byte-count streaming kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM:
the counters are calibrated only for 16 B/lane streams; the conv kernels here issue 4 B/lane loads).  Three launches each of
  k_add_srcs  (sc_apply_src, RAW source): 4 B/lane coalesced loads and stores, 1 GiB in, 1 GiB out (4x the 256 MiB Infinity Cache)
  torch copy_ : 16 B/lane loads and stores, 1 GiB in, 1 GiB out"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import _lib
from starcop_amd._lib import SRC_RAW, check, make_src, ptr, stream
lib = _lib.load()
N, Cc, HW = 16, 64, 512 * 512                      # 2^28 floats = 1 GiB
x = torch.randn(N, Cc, HW, device="cuda")
y = torch.empty_like(x)
s = make_src(x, Cc, SRC_RAW)
for _ in range(3):
    check(lib.sc_apply_src(C.byref(s), ptr(y), N, Cc, HW, stream()))
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("bytes per launch:", x.numel() * 4)
