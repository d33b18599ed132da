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
names = {7: "prologue", 8: "band means", 9: "covariance C_0", 10: "(alpha = 0: inverse)", 12: "init", 4: "it: C_k build", 5: "it: blocked Cholesky", 0: "(alpha = 0: W [v t])",
         1: "it: solve by wave 0 + dots", 2: "it: the pass over X", 3: "it: band sums, v, next t"}
for k in (7, 8, 9, 10, 12, 4, 5, 0, 1, 2, 3):
    print(f"{names[k]:32s} {d[k]:9.1f} us" + (f"   ({d[k] / 31:.2f} us per iteration)" if k < 7 else ""))
print(f"{'group total':32s} {d[:13].sum():9.1f} us")
