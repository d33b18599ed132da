"""Evaluation-mask kernels: GB/s of the 16-threshold PR sweep with and without the 3x3 opening, and the
numpy oracle beside it.  Algorithmic bytes: prediction + label read once = 8 B per pixel (int64 mask write: +8 B)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import host_ref  # noqa: E402  (cpu leg only)
from starcop_amd import baselines, validation  # noqa: E402
from starcop_amd._lib import SE_CROSS  # noqa: E402

N, H, W = 64, 512, 512
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.rand((N, 1, H, W), device="cuda", generator=g)
y = (torch.rand((N, 1, H, W), device="cuda", generator=g) < 0.2).float()
thr = np.sort([0, 1e-3, 1e-2] + np.arange(0.5, .96, .05).tolist() + [.99, .995, .999])[::-1]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, bits in (("plain", 0), ("cross opening", SE_CROSS)):
    cm = torch.zeros((N, len(thr), 2, 2), dtype=torch.int64, device="cuda")
    ms = timeit(lambda: validation.threshold_confusion(p, y, thr, bits, out=cm))
    print(f"PR sweep T=16 {name:14s}: {ms:.3f} ms for {N} tiles  {N / ms * 1e3:8.0f} tiles/s  {N * H * W * 8 / ms / 1e6:7.1f} GB/s")
    ms = timeit(lambda: baselines.thresholded_opening(p, 0.5, bits))
    print(f"mask (int64 out) {name:14s}: {ms:.3f} ms  {N * H * W * 12 / ms / 1e6:7.1f} GB/s")
pc, yc = p[0, 0].cpu().numpy(), y[0, 0].cpu().numpy()
t0 = time.perf_counter()
for t in thr:
    host_ref.confusion(host_ref.apply_threshold(pc, t, np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])), yc)
print(f"numpy oracle, one tile, 16 thresholds with opening: {(time.perf_counter() - t0) * 1e3:.1f} ms")
