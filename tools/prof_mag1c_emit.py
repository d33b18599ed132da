"""Phase times of one group of the EMIT case (k_mag1c: alpha = 1e-4, 1280 x 1242 x 49, column_step 2 -> 621 groups of 2560 pixels)
from the -DSTARCOP_MAG1C_PROF build (tools/prof_mag1c_phases.py --build)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["STARCOP_HIP_LIB"] = os.path.join(ROOT, "tools", "_build", os.environ.get("PROF_LIB", "libstarcop_hip_prof.so"))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from starcop_amd import mag1c, _lib   # noqa: E402

lib = _lib.load()
g3 = np.load(os.path.join(ROOT, "tests", "golden", "g3_templates.npz"))
te = g3["emit_template_kept"][:, 1]
rng = np.random.default_rng(0)
base = rng.uniform(1, 6, size=te.size)
raw = torch.from_numpy((base * (1 + 0.05 * rng.standard_normal((1280, 1242, te.size)))).astype(np.float32)).cuda()
buf = (ctypes.c_longlong * 32)()


def snap():
    torch.cuda.synchronize()
    lib.sc_debug_mag1c_prof(buf)
    return np.array(list(buf), dtype=np.int64)


mag1c.mag1c_columns(raw, te, -9999.0, column_step=2)
b0 = snap()
mag1c.mag1c_columns(raw, te, -9999.0, column_step=2)
d = (snap() - b0) / 100.0
names = {16: "band means", 17: "covariance C_0", 18: "it: mu, t, C_k", 19: "it: Cholesky", 20: "it: two solves + dots", 21: "it: pixel sweep",
         22: "it: block sums", 23: "it: v = X^T w"}
for k in range(16, 24):
    print(f"{names[k]:24s} {d[k]:9.1f} us" + (f"   ({d[k] / 31:.2f} us per iteration)" if k >= 18 else ""))
print(f"{'group total':24s} {d[16:24].sum():9.1f} us")
