"""Inference throughput: eval-mode forward + masks (batch_with_preds semantics) on 512x512 tiles, and predict() on an
EMIT-sized scene (1280x1242 -> reflect-padded to 1280x1248)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_batch
from starcop_amd import model_module as mm

torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()
B = 16
batch = synth_batch(B, 512, 512, 1, dev)


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    dt = timeit(lambda: model(batch["input"]), 10)
    print(f"eval forward B={B}: {dt*1e3:.2f} ms  {B/dt:.0f} tiles/s")
    dt = timeit(lambda: model.batch_with_preds(batch), 10)
    print(f"batch_with_preds (forward + sigmoid + masks + loss_per_pixel) B={B}: {dt*1e3:.2f} ms  {B/dt:.0f} tiles/s")
scene = synth_batch(1, 1280, 1248, 2, "cpu")["input"][0, :, :, :1242].numpy()
dt = timeit(lambda: model.predict(scene), 3)
print(f"predict() on a 4x1280x1242 scene (H2D + pad + forward + D2H): {dt*1e3:.1f} ms  ({1280*1242/262144/dt:.0f} tile-eq/s)")
