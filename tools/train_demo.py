"""End-to-end on synthetic data with every stage on the GPU: tiles resident in HBM -> weighted 128x128 crops with rotation / flips
(starcop_amd.datamodule) -> fused training steps (ModelModule.fused_train_step) -> run_validation on the full tiles, with the
Mag1cBaseline beside it.  usage: python tools/train_demo.py [n_tiles=32] [tile=256] [epochs=3] [batch=32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starcop_amd import baselines, datamodule as dm, validation  # noqa: E402
from starcop_amd.model_module import ModelModule, default_settings  # noqa: E402


def synth_tiles(n, T, seed, device):
    """mag1c-like channel (noise + Gaussian plumes) + smooth RGB; label = the planted plume (every third tile has none, every third a
    large one > 1000 px, every third a small one: the three groups run_validation reports on)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(T).float(), torch.arange(T).float(), indexing="ij")
    mag = (torch.randn(n, 1, T, T, generator=g) * 120).abs()
    lab = torch.zeros(n, 1, T, T)
    for i in range(n):
        if i % 3:
            sig = 28.0 if i % 3 == 1 else 7.0
            cy, cx = (float(torch.rand(1, generator=g)) * 0.6 + 0.2) * T, (float(torch.rand(1, generator=g)) * 0.6 + 0.2) * T
            blob = 1800 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig ** 2))
            mag[i, 0] += blob
            lab[i, 0] = (blob > 450).float()
    rgb = 57.5 + 13.0 * torch.cos(yy[None, None] * 6.28 / T + torch.rand(n, 3, 1, 1, generator=g) * 6.28) * torch.cos(xx[None, None] * 12.56 / T)
    x = torch.cat([mag, rgb], 1).float()
    return {"input": x.to(device), "output": lab.to(device), "weight_loss": (mag / 400).clamp(0.1, 1).to(device)}


def main(n_tiles=32, tile=256, epochs=3, batch=32, lr=1e-3, quiet=False):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    data = synth_tiles(n_tiles, tile, 7, dev)
    ts = dm.ResidentTileSet(data["input"], data["output"], data["weight_loss"], device=dev)
    loader = dm.TrainLoader(ts, batch_size=batch, training_size=(128, 128), seed=1)
    model = ModelModule(default_settings(pos_weight=1, lr=lr)).to(dev).train()
    opt = model.configure_optimizers()["optimizer"]
    hist = []
    for ep in range(epochs):
        t0, acc, n = time.perf_counter(), 0.0, 0
        for b in loader:
            acc_dev = model.fused_train_step(b, opt)
            acc += float(acc_dev) / b["output"].numel(); n += 1
        torch.cuda.synchronize()
        hist.append(acc / n)
        if not quiet:
            print(f"epoch {ep}: mean loss {acc / n:.4f}  {n * batch / (time.perf_counter() - t0):.0f} crops/s")
    val = [{"input": data["input"][i:i + 1], "output": data["output"][i:i + 1], "weight_loss": data["weight_loss"][i:i + 1],
            "id": [ts.ids[i]], "has_plume": torch.tensor([int(data["output"][i].sum() > 0)])} for i in range(n_tiles)]
    with np.errstate(all="ignore"):
        df, met = validation.run_validation(model, val, verbose=False)
        _, met_b = validation.run_validation(baselines.Mag1cBaseline(default_settings().dataset.input_products).to(dev), val, verbose=False)
    if not quiet:
        print(f"U-Net   : F1 {met['f1score']:.3f}  IoU {met['iou']:.3f}  tile-classification F1 {met['classification_f1score']:.3f}")
        print(f"baseline: F1 {met_b['f1score']:.3f}  IoU {met_b['iou']:.3f}  (mag1c > 500 + 3x3 opening)")
    return hist, met, met_b


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]]
    main(*a)
