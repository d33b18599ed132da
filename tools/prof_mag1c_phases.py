"""Where the time of one group goes inside k_mag1c_res (configs[2]: 512 groups x 512 px x 125 bands, 30 iterations).

The kernel carries wall-clock probes behind -DSTARCOP_MAG1C_PROF (thread 0 of group 0 accumulates the 100 MHz counter between
phase boundaries).  `--build` (no GPU needed) compiles that variant of mag1c.hip and links it with the product objects into
tools/_build/libstarcop_hip_prof.so; without flags the script runs on the GPU box against that library.

  python tools/prof_mag1c_phases.py --build && gpurun -- python tools/prof_mag1c_phases.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build")
LIB = os.path.join(OUT, os.environ.get("PROF_LIB", "libstarcop_hip_prof.so"))
CSRC = os.path.join(ROOT, "starcop_amd", "csrc")

if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["make", "-C", CSRC, "-j", "4"], check=True, stdout=subprocess.DEVNULL)
    obj = os.path.join(OUT, "mag1c_prof.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, "-Wno-unused-result", "-DSTARCOP_MAG1C_PROF"] + os.environ.get("EXTRA_DEFS", "").split() + ["-c", os.path.join(CSRC, "mag1c.hip"), "-o", obj], check=True)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "mag1c.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", LIB], check=True)
    print("built", LIB)
    sys.exit(0)

os.environ["STARCOP_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from starcop_amd import mag1c, _lib   # noqa: E402

lib = _lib.load()
S = int(os.environ.get("S", "125"))
rng = np.random.default_rng(0)
base = rng.uniform(1, 6, size=S)
x = torch.from_numpy((base * (1 + 0.05 * rng.standard_normal((512, 512, S)))).astype(np.float32)).cuda()
t = rng.uniform(-1, 0, size=S)
groups = np.arange(1, 513)[None, :].repeat(512, 0)
buf = (ctypes.c_longlong * 32)()


def snap():
    torch.cuda.synchronize()
    lib.sc_debug_mag1c_prof(buf)
    return np.array(list(buf), dtype=np.int64)


mag1c.acrwl1mf_by_groups(x, t, groups)
b0 = snap()
mag1c.acrwl1mf_by_groups(x, t, groups)
d = (snap() - b0) / 100.0          # us
names = {7: "tile load", 8: "band means", 9: "covariance", 10: "Cholesky", 11: "L^-1", 12: "W = X^T X", 0: "it: W [v t] partials", 1: "it: sums + ten dots",
         2: "it: 2x2 + pixel sweep", 3: "it: v = X^T w"}
for k in (7, 8, 9, 10, 11, 12, 0, 1, 2, 3):
    print(f"{names[k]:24s} {d[k]:8.1f} us" + (f"   ({d[k] / 31:.2f} us per iteration)" if k < 7 else ""))
print(f"{'setup':24s} {d[7:13].sum():8.1f} us\n{'iterations':24s} {d[:7].sum():8.1f} us\n{'group total':24s} {d.sum():8.1f} us")
