"""Training data path with the tiles resident in HBM.

Mirrors the sample-selection and augmentation logic of /root/reference/starcop/data/datamodule.py and dataset.py:
  ``create_windows``          georeader.slices.create_windows as called at datamodule.py:27-28 (third-party, absent: restated
                              for the one call the reference makes -- (512,512) tiles, window 128, overlap 64, complete windows)
  ``tiled_table``             ``tiled_dataframe`` :17-64: one row per window, ``frac_positives``, ``has_plume`` = frac > 10/64**2,
                              id ``{id}_r{row}_c{col}_w{w}_h{h}``
  ``add_sample_weight``       :342-348 (1/plume_fraction vs 1/(1-plume_fraction))
  ``TrainLoader``             ``train_dataloader`` :306-326: ``WeightedRandomSampler(weights, num_samples=len, replacement=True)``
                              (= ``torch.multinomial``), batches of dict(input, output, weight_loss, id, has_plume)
  augmentation                :128-134 kornia ``RandomRotation(p=.5, degrees=90)`` -> ``RandomHorizontalFlip(p=.5)`` ->
                              ``RandomVerticalFlip(p=.5)`` applied to input, label and loss weight with the same parameters
                              (dataset.py:99-102)

What is different underneath: the reference decodes one GeoTIFF window per product per sample in DataLoader workers and
augments on the CPU; 1 GPU consumes ~940 tiles/s = 23 GB/s of decoded fp32 samples, which no host pipeline sustains.  The
whole STARCOP training set (~3 400 tiles x 6 products x 1 MB = 20 GB) fits 14 times into one MI355X's 288 GB, so the tiles
are uploaded once and every batch is cut, rotated and flipped by ONE gather kernel per tensor (``sc_gather_augment``)
straight out of HBM.  The sample folders on disk (one tiled GeoTIFF per product) are decoded by ``io_formats.load_tileset``
(own TIFF reader, thread pool, pinned staging buffers, asynchronous upload) into the ``ResidentTileSet`` below.
kornia is absent from the build image: the rotation follows kornia 0.6.7's ``rotate`` -> ``warp_affine`` ->
``F.grid_sample(align_corners=True, padding_mode="zeros")`` chain and is tested against ``F.grid_sample`` itself.
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch

from . import _lib
from ._lib import check, ptr, stream

ROTATE, HFLIP, VFLIP = 1, 2, 4


def create_windows(shape: Tuple[int, int], window_size: Tuple[int, int], overlap: Tuple[int, int],
                   include_incomplete: bool = False) -> List[Tuple[int, int, int, int]]:
    """(row_off, col_off, height, width) windows on a ``shape`` grid, row-major, stride = window - overlap."""
    sr, sc = window_size[0] - overlap[0], window_size[1] - overlap[1]
    assert sr > 0 and sc > 0, "overlap must be smaller than the window"
    out = []
    for r in range(0, shape[0], sr):
        for c in range(0, shape[1], sc):
            h, w = min(window_size[0], shape[0] - r), min(window_size[1], shape[1] - c)
            if (h, w) != tuple(window_size) and not include_incomplete:
                continue
            out.append((r, c, h, w))
    return out


def add_sample_weight(dataframe: pd.DataFrame) -> pd.DataFrame:
    plume_fraction = np.sum(dataframe["has_plume"]) / dataframe.shape[0]
    plume_weight = 1 / plume_fraction
    non_plume_weight = 1 / (1 - plume_fraction)
    dataframe["sample_weight"] = dataframe["has_plume"].apply(lambda x: plume_weight if x else non_plume_weight)
    return dataframe


def gather_augment(tiles: torch.Tensor, tile, row_off, col_off, cos_t, sin_t, flags, size: Tuple[int, int],
                   mode: str = "bilinear") -> torch.Tensor:
    """(B,C,h,w) batch cut from ``tiles`` (M,C,H,W) on the device; per-item int32 / float32 device tensors."""
    _lib.require_device(tiles)
    lib = _lib.load()
    assert tiles.dtype == torch.float32 and tiles.is_contiguous() and tiles.dim() == 4
    M, C_, Hs, Ws = tiles.shape
    B = tile.numel()
    out = torch.empty((B, C_, size[0], size[1]), dtype=torch.float32, device=tiles.device)
    check(lib.sc_gather_augment(ptr(tiles), M, C_, Hs, Ws, ptr(tile), ptr(row_off), ptr(col_off), ptr(cos_t), ptr(sin_t),
                                ptr(flags), B, size[0], size[1], {"bilinear": 0, "nearest": 1}[mode], ptr(out), stream()))
    return out


class ResidentTileSet:
    """All samples of a split on the device: ``inputs`` (M,C,H,W), ``outputs`` (M,1,H,W), optional ``weight_loss`` (M,1,H,W)."""

    def __init__(self, inputs, outputs, weight_loss=None, ids: Optional[Sequence[str]] = None, device="cuda"):
        def up(t):
            return None if t is None else torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()
        self.inputs, self.outputs, self.weight_loss = up(inputs), up(outputs), up(weight_loss)
        M = self.inputs.shape[0]
        assert self.outputs.shape[0] == M and self.outputs.shape[-2:] == self.inputs.shape[-2:]
        self.ids = list(ids) if ids is not None else [f"sample_{i:05d}" for i in range(M)]
        self.shape = tuple(self.inputs.shape[-2:])

    def __len__(self):
        return self.inputs.shape[0]

    def tiled_table(self, tile_size=(128, 128), overlap=(64, 64)) -> pd.DataFrame:
        """One row per training window (datamodule.py:17-64); the label fractions are summed on the device."""
        wins = create_windows(self.shape, tile_size, overlap, include_incomplete=False)
        lab = self.outputs[:, 0]
        # integral image per tile -> window sums with four look-ups (exact for {0,1} labels: sums < 2^24 stay exact in f64)
        ii = torch.zeros((lab.shape[0], lab.shape[1] + 1, lab.shape[2] + 1), dtype=torch.float64, device=lab.device)
        ii[:, 1:, 1:] = lab.double().cumsum(1).cumsum(2)
        r = torch.tensor([w[0] for w in wins], device=lab.device)
        c = torch.tensor([w[1] for w in wins], device=lab.device)
        h, w_ = tile_size
        sums = (ii[:, r + h][:, torch.arange(len(wins)), c + w_] - ii[:, r][:, torch.arange(len(wins)), c + w_]
                - ii[:, r + h][:, torch.arange(len(wins)), c] + ii[:, r][:, torch.arange(len(wins)), c]).cpu().numpy()
        rows = []
        for m, sid in enumerate(self.ids):
            for k, (ro, co, hh, ww) in enumerate(wins):
                rows.append({"id": f"{sid}_r{ro}_c{co}_w{ww}_h{hh}", "id_original": sid, "tile": m, "window_row_off": ro,
                             "window_col_off": co, "window_width": ww, "window_height": hh,
                             "frac_positives": sums[m, k] / (hh * ww)})
        df = pd.DataFrame(rows)
        df["has_plume"] = df["frac_positives"] > (10 / 64 ** 2)
        return df.set_index("id")


class TrainLoader:
    """Iterable of training batches drawn like the reference's ``train_dataloader`` and augmented on the device."""

    def __init__(self, tileset: ResidentTileSet, table: Optional[pd.DataFrame] = None, batch_size: int = 32,
                 training_size=(128, 128), weight_sampling: bool = True, augment: bool = True, seed: int = 0,
                 mask_mode: str = "bilinear", drop_last: bool = False):
        self.ts = tileset
        self.size = tuple(training_size)
        if table is None:
            if self.size == tileset.shape:
                lab = tileset.outputs.flatten(1)
                table = pd.DataFrame({"id": tileset.ids, "tile": np.arange(len(tileset)), "window_row_off": 0, "window_col_off": 0,
                                      "frac_positives": (lab.sum(1) / lab.shape[1]).cpu().numpy()})
                table["has_plume"] = table["frac_positives"] > (10 / 64 ** 2)
                table = table.set_index("id")
            else:
                table = tileset.tiled_table(self.size)
        self.table = table
        self.batch_size, self.weight_sampling, self.augment = batch_size, weight_sampling, augment
        self.mask_mode, self.drop_last = mask_mode, drop_last
        self.gen = torch.Generator().manual_seed(seed)
        dev = tileset.inputs.device
        self._tile = torch.as_tensor(table["tile"].values, dtype=torch.int32)
        self._row = torch.as_tensor(table["window_row_off"].values, dtype=torch.int32)
        self._col = torch.as_tensor(table["window_col_off"].values, dtype=torch.int32)
        self._has = torch.as_tensor(table["has_plume"].values.astype(np.int64))
        self._ids = list(table.index)
        self._dev = dev

    def __len__(self):
        n = len(self.table)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def epoch_indices(self) -> torch.Tensor:
        """Sample order of one epoch (datamodule.py:311-326)."""
        n = len(self.table)
        if self.weight_sampling:
            w = torch.as_tensor(add_sample_weight(self.table.copy())["sample_weight"].values, dtype=torch.double)
            return torch.multinomial(w, n, True, generator=self.gen)          # == WeightedRandomSampler(w, n, replacement=True)
        return torch.randperm(n, generator=self.gen)

    def draw_augmentation(self, n: int):
        """(cos, sin, flags) of n samples: rotation by U(-90, 90) degrees w.p. .5, h-flip w.p. .5, v-flip w.p. .5."""
        if not self.augment:
            return torch.ones(n), torch.zeros(n), torch.zeros(n, dtype=torch.int32)
        u = torch.rand((4, n), generator=self.gen)
        ang = (u[1] * 180.0 - 90.0) * (math.pi / 180.0)
        rot = u[0] < 0.5
        flags = rot.int() * ROTATE + (u[2] < 0.5).int() * HFLIP + (u[3] < 0.5).int() * VFLIP
        return torch.where(rot, torch.cos(ang), torch.ones(n)), torch.where(rot, torch.sin(ang), torch.zeros(n)), flags.int()

    def make_batch(self, idx: torch.Tensor, cos_t, sin_t, flags):
        dev = self._dev
        a = [t.to(dev, non_blocking=True) for t in (self._tile[idx], self._row[idx], self._col[idx], cos_t.float(), sin_t.float(), flags)]
        ts = self.ts
        batch = {"input": gather_augment(ts.inputs, *a, self.size),
                 "output": gather_augment(ts.outputs, *a, self.size, mode=self.mask_mode)}
        if ts.weight_loss is not None:
            batch["weight_loss"] = gather_augment(ts.weight_loss, *a, self.size)
        batch["id"] = [self._ids[i] for i in idx.tolist()]
        batch["has_plume"] = self._has[idx].to(dev)
        return batch

    def __iter__(self):
        order = self.epoch_indices()
        for s in range(0, order.numel(), self.batch_size):
            idx = order[s:s + self.batch_size]
            if self.drop_last and idx.numel() < self.batch_size:
                break
            yield self.make_batch(idx, *self.draw_augmentation(idx.numel()))
