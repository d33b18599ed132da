"""Batches/s of the HBM-resident training loader (crop + rotate + flip gather) vs what one train step consumes."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from starcop_amd import datamodule as dm  # noqa: E402

M = 64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((M, 4, 512, 512), device="cuda", generator=g) * 100
y = (torch.rand((M, 1, 512, 512), device="cuda", generator=g) < 0.01).float()
ts = dm.ResidentTileSet(x, y, torch.clamp(x[:, :1] / 400, 0.1, 1), device="cuda")
for size, bs in (((128, 128), 32), ((512, 512), 16)):
    loader = dm.TrainLoader(ts, batch_size=bs, training_size=size, seed=1)
    n = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for ep in range(3 if size[0] == 128 else 20):
        for b in loader:
            n += b["input"].shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    px = n * size[0] * size[1]
    print(f"{size[0]}x{size[1]} batch {bs}: {n / dt:9.0f} samples/s = {px / dt / 512 / 512:8.0f} tile-equivalents/s "
          f"({px * 6 * 4 * 2 / dt / 1e9:.1f} GB/s gathered+written)")
