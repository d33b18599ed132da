"""HBM-resident training data path: crop + rotation + flips gather kernel against the oracle (exact for
copies and flips, 1e-5 for the bilinear rotation), the tiled sample table, the weighted sampler and a short training run."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import host_ref  # noqa: E402
from starcop_amd import datamodule as dm  # noqa: E402

DEV = "cuda"


def _items(tile, row, col, deg, flags):
    a = np.deg2rad(np.asarray(deg, np.float64))
    t = lambda v, d: torch.tensor(v, dtype=d, device=DEV)     # noqa: E731
    return (t(tile, torch.int32), t(row, torch.int32), t(col, torch.int32), t(np.cos(a), torch.float32),
            t(np.sin(a), torch.float32), t(flags, torch.int32))


def test_gather_augment_matches_oracle(hip):
    rng = np.random.default_rng(0)
    tiles = rng.normal(size=(3, 4, 96, 80)).astype(np.float32)
    td = torch.from_numpy(tiles).to(DEV)
    tile = [0, 2, 1, 1, 0, 2, 2, 0]
    row = [0, 10, 32, 5, 64, 0, 17, 3]
    col = [0, 3, 16, 40, 48, 7, 1, 9]
    deg = [0.0, 30.0, -90.0, 90.0, 45.0, -12.5, 77.0, 0.0]
    flags = [0, 1, 1 | 2, 1 | 4, 2 | 4, 1 | 2 | 4, 1, 4]
    for (h, w) in [(32, 32), (31, 17)]:
        for mode in ("bilinear", "nearest"):
            got = dm.gather_augment(td, *_items(tile, row, col, deg, flags), (h, w), mode=mode).cpu().numpy()
            assert got.shape == (8, 4, h, w)
            for b in range(8):
                crop = tiles[tile[b], :, row[b]:row[b] + h, col[b]:col[b] + w]
                a = math.radians(deg[b])
                want = host_ref.rotate_flip(crop, np.float32(math.cos(a)), np.float32(math.sin(a)), flags[b] & 1, flags[b] & 2,
                                            flags[b] & 4, nearest=(mode == "nearest"))
                if not flags[b] & 1:
                    assert np.array_equal(got[b], want), (b, mode)            # copies and flips are exact
                elif mode == "bilinear":
                    assert np.abs(got[b] - want).max() < 1e-5, (b, np.abs(got[b] - want).max())
                else:
                    assert (got[b] != want).mean() < 5e-3                     # rounding ties only
    with pytest.raises(ValueError):
        dm.gather_augment(td, *_items([0], [0], [0], [0.0], [0]), (128, 32))


def _tileset(rng, M=5, H=256, W=256):
    x = rng.uniform(0, 100, size=(M, 4, H, W)).astype(np.float32)
    y = np.zeros((M, 1, H, W), np.float32)
    y[0, 0, 10:90, 20:100] = 1
    y[2, 0, 200:203, 100:104] = 1               # 12 px: exceeds the 10/64^2 fraction only in... no window (12/16384 < 10/4096)
    y[3, 0, 128:192, 64:128] = 1
    wl = np.clip(x[:, :1] / 400, 0.1, 1).astype(np.float32)
    return dm.ResidentTileSet(x, y, wl, ids=[f"ang{i}" for i in range(M)], device=DEV), x, y, wl


def test_tiled_table_and_loader(hip):
    rng = np.random.default_rng(4)
    ts, x, y, wl = _tileset(rng)
    table = ts.tiled_table((128, 128), (64, 64))
    wins = dm.create_windows((256, 256), (128, 128), (64, 64))
    assert len(wins) == 9 and len(table) == 5 * 9
    for (sid, m) in (("ang0", 0), ("ang3", 3), ("ang2", 2)):
        for (r, c, h, w) in wins:
            rowd = table.loc[f"{sid}_r{r}_c{c}_w{w}_h{h}"]
            frac = y[m, 0, r:r + h, c:c + w].sum() / (h * w)
            assert rowd["frac_positives"] == frac and bool(rowd["has_plume"]) == (frac > 10 / 64 ** 2)
            assert rowd["tile"] == m and rowd["id_original"] == sid
    assert table["has_plume"].sum() > 0 and not table.loc[table["id_original"] == "ang2", "has_plume"].any()

    loader = dm.TrainLoader(ts, table, batch_size=8, training_size=(128, 128), weight_sampling=True, augment=False, seed=7)
    assert len(loader) == 6
    # the sample order is WeightedRandomSampler's: torch.multinomial on the same weights and generator state
    w = torch.as_tensor(dm.add_sample_weight(table.copy())["sample_weight"].values, dtype=torch.double)
    want_order = torch.multinomial(w, len(table), True, generator=torch.Generator().manual_seed(7))
    seen, plume = [], 0
    for k, batch in enumerate(loader):
        idx = want_order[k * 8:(k + 1) * 8]
        assert batch["id"] == [table.index[i] for i in idx.tolist()]
        assert batch["input"].shape == (len(idx), 4, 128, 128) and batch["output"].shape == (len(idx), 1, 128, 128)
        for j, i in enumerate(idx.tolist()):
            rowd = table.iloc[i]
            m, r, c = int(rowd["tile"]), int(rowd["window_row_off"]), int(rowd["window_col_off"])
            assert np.array_equal(batch["input"][j].cpu().numpy(), x[m, :, r:r + 128, c:c + 128])
            assert np.array_equal(batch["output"][j].cpu().numpy(), y[m, :, r:r + 128, c:c + 128])
            assert np.array_equal(batch["weight_loss"][j].cpu().numpy(), wl[m, :, r:r + 128, c:c + 128])
            assert int(batch["has_plume"][j]) == int(rowd["has_plume"])
        seen += idx.tolist(); plume += int(batch["has_plume"].sum())
    assert len(seen) == 45 and 0.25 < plume / 45 < 0.75          # plume / no-plume windows balanced by the weights


def test_augmented_batches_keep_input_label_weight_aligned_and_train(hip):
    from starcop_amd.model_module import ModelModule, default_settings
    rng = np.random.default_rng(9)
    ts, x, y, wl = _tileset(rng)
    loader = dm.TrainLoader(ts, batch_size=8, training_size=(128, 128), augment=True, seed=3)
    g = torch.Generator().manual_seed(3)
    order = torch.multinomial(torch.as_tensor(dm.add_sample_weight(loader.table.copy())["sample_weight"].values, dtype=torch.double),
                              len(loader.table), True, generator=g)
    batch = next(iter(loader))
    u = torch.rand((4, 8), generator=g)
    assert batch["input"].shape == (8, 4, 128, 128)
    kinds = set()
    for j, i in enumerate(order[:8].tolist()):
        rowd = loader.table.iloc[i]
        m, r, c = int(rowd["tile"]), int(rowd["window_row_off"]), int(rowd["window_col_off"])
        rot, hf, vf = bool(u[0, j] < 0.5), bool(u[2, j] < 0.5), bool(u[3, j] < 0.5)
        a = math.radians(float(u[1, j]) * 180.0 - 90.0)
        cs, sn = (np.float32(math.cos(a)), np.float32(math.sin(a))) if rot else (np.float32(1), np.float32(0))
        kinds.add((rot, hf, vf))
        for key, src in (("input", x), ("output", y), ("weight_loss", wl)):
            want = host_ref.rotate_flip(src[m, :, r:r + 128, c:c + 128], cs, sn, rot, hf, vf)
            assert np.abs(batch[key][j].cpu().numpy() - want).max() < 2e-4 * max(1.0, float(np.abs(want).max())), (j, key)
    assert len(kinds) > 2
    # a few optimiser steps straight from the loader (the reference's 128x128 training recipe)
    torch.manual_seed(0)
    model = ModelModule(default_settings()).to(DEV).train()
    losses = []
    for k, b in enumerate(loader):
        acc = model.fused_train_step(b)
        losses.append(float(acc) / b["output"].numel())
        if k == 3:
            break
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 1.5


def test_end_to_end_train_then_validate(hip):
    """tools/train_demo.py at a small size: resident tiles -> augmented crops -> fused train steps -> run_validation of the
    network and of the mag1c baseline; the loss must fall and the metrics must be well-formed."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("train_demo", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            "tools", "train_demo.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    hist, met, met_b = mod.main(n_tiles=12, tile=256, epochs=3, batch=16, quiet=True)
    assert hist[-1] < hist[0] and all(np.isfinite(hist))
    assert 0.0 <= met["iou"] <= 1.0 and int(met["confusion_matrix"].sum()) == 12 * 256 * 256
    assert met_b["f1score"] > 0.8                       # the label is the planted plume above ~450: mag1c > 500 nearly recovers it


def test_disk_samples_to_resident_tiles_to_train_step(hip, tmp_path):
    """f2 end to end: sample folders on disk in the reference's layout (one tiled GeoTIFF per product, dataset.py:59-102) ->
    io_formats.load_tileset (pinned staging, async upload) -> ResidentTileSet -> TrainLoader crops -> fused train steps; and the
    un-augmented batch equals what the reference's Dataset.__getitem__ assembles from the same files (read back with read_tiff)."""
    from starcop_amd import io_formats as io, model_module as mm
    rng = np.random.default_rng(0)
    prods = ["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"]
    folders = []
    for i in range(3):
        d = tmp_path / f"ang2019_{i:03d}"
        d.mkdir()
        mag = np.clip(rng.normal(0, 400, (512, 512)), 0, None).astype(np.float32)
        mag[100:160, 200 + 20 * i:260 + 20 * i] += 1500
        io.write_tiff(str(d / "mag1c.tif"), mag)
        for p in prods[1:]:
            io.write_tiff(str(d / f"{p}.tif"), rng.uniform(5, 110, (512, 512)).astype(np.float32), compress=None)
        io.write_tiff(str(d / "labelbinary.tif"), (mag > 900).astype(np.uint8))
        io.write_tiff(str(d / "weight_mag1c.tif"), np.clip(mag / 400, 0.1, 1).astype(np.float32))
        folders.append(str(d))
    ts = io.load_tileset(folders, prods, ("labelbinary",), "weight_mag1c", device=DEV)
    assert len(ts) == 3 and ts.inputs.shape == (3, 4, 512, 512) and ts.ids[1] == "ang2019_001"
    for i, f in enumerate(folders):
        assert np.array_equal(ts.inputs[i].cpu().numpy(), io.load_sample(f, prods))
        assert np.array_equal(ts.outputs[i, 0].cpu().numpy(), io.read_tiff(os.path.join(f, "labelbinary.tif"))[0].astype(np.float32))
    loader = dm.TrainLoader(ts, batch_size=8, training_size=(128, 128), augment=False, weight_sampling=False, seed=1)
    batch = next(iter(loader))
    row = loader.table.loc[batch["id"][0]]
    win = (int(row.window_row_off), int(row.window_col_off), 128, 128)
    assert np.array_equal(batch["input"][0].cpu().numpy(), io.load_sample(folders[int(row.tile)], prods, win))
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(DEV).train()
    opt = model.configure_optimizers()["optimizer"]
    aug = dm.TrainLoader(ts, batch_size=8, training_size=(128, 128), seed=2)
    losses = [float(model.fused_train_step(b, opt).item()) / (8 * 128 * 128) for b, _ in zip(aug, range(4))]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 1.5
