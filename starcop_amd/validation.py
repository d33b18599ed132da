"""``run_validation``: per-tile and aggregated segmentation / classification metrics of a model over a dataloader.

Mirrors /root/reference/starcop/validation.py:26-224 (same signature, same DataFrame columns, same keys in the returned
metrics dict, same ``results.csv`` / ``results_agg.json`` files) and ``to_device`` of starcop/torch_utils.py:5-12.

What is different underneath: the reference updates 1 + T ``torchmetrics.ConfusionMatrix`` objects per tile (T = 16 PR
thresholds by default, each a threshold pass + a bincount, and for the morphological baselines two kornia unfold passes)
and reads ~10 scalars back per tile.  Here the T thresholded (and opened) masks are counted against the label in ONE pass
over the prediction (``sc_threshold_confusion``), the per-tile matrices stay on the device, and there is a single
read-back at the end of the loop.  Plotting (``products_plot``) is outside the hot path and is skipped with a warning.
"""
import json
import os
import tempfile
import warnings
from numbers import Number
from typing import Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from . import _lib, metrics as starcopmetrics
from ._lib import check, ptr, stream

MAX_T = 32          # thresholds per sc_threshold_confusion launch


def to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device)
    if hasattr(x, "keys"):
        return {k: to_device(v, device) for k, v in x.items()}
    return x


def threshold_confusion(pred: torch.Tensor, target: torch.Tensor, thresholds, se_bits: int = 0, ignore=None,
                        out: Optional[torch.Tensor] = None, invalid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N, T, 2, 2) int64 confusion matrices ``cm[n, t, target, prediction]`` of ``opening(pred > thresholds[t])`` against
    ``target.long()`` for N tiles ((N,H,W) or (N,1,H,W) float tensors); ``out`` is accumulated into when given."""
    _lib.require_device(pred)
    lib = _lib.load()
    p = pred.contiguous().float()
    t = target.contiguous().float()
    assert p.shape == t.shape, f"prediction {tuple(p.shape)} vs target {tuple(t.shape)}"
    H, W = p.shape[-2:]
    n = p.numel() // (H * W)
    thr = np.ascontiguousarray(np.asarray(thresholds, dtype=np.float32).reshape(-1))
    T = thr.shape[0]
    cm = out if out is not None else torch.zeros((n, T, 2, 2), dtype=torch.int64, device=p.device)
    assert cm.shape == (n, T, 2, 2) and cm.dtype == torch.int64 and cm.is_contiguous()
    ig = None
    if ignore is not None:
        ig = (ignore != 0).contiguous().to(torch.uint8)
        assert ig.numel() == p.numel()
    if T <= MAX_T:
        check(lib.sc_threshold_confusion(ptr(p), ptr(t), ptr(ig), thr.ctypes.data, T, int(se_bits), ptr(cm), ptr(invalid),
                                         n, H, W, stream()))
    else:       # chunks write strided slices: go through a temporary
        for a in range(0, T, MAX_T):
            b = min(T, a + MAX_T)
            part = torch.zeros((n, b - a, 2, 2), dtype=torch.int64, device=p.device)
            check(lib.sc_threshold_confusion(ptr(p), ptr(t), ptr(ig), thr[a:b].ctypes.data, b - a, int(se_bits), ptr(part),
                                             ptr(invalid) if a == 0 else None, n, H, W, stream()))
            cm[:, a:b] += part
    return cm


@torch.no_grad()
def run_validation(model, dataloader, products_plot: Optional[List[str]] = None, verbose: bool = True,
                   thresholds=None, show_plots: bool = True, path_save_results: Optional[str] = None,
                   skip_saving_plots=False, mask_from_magic=False) -> Tuple[pd.DataFrame, Dict[str, Number]]:
    assert getattr(dataloader, "batch_size", 1) == 1, "This function is expected to run with batch_size 1"
    if thresholds is None:
        thresholds = [0, 1e-3, 1e-2] + np.arange(0.5, .96, .05).tolist() + [.99, .995, .999]
    thresholds = np.sort(thresholds)[-1::-1]                     # high to low (validation.py:41-42)
    T = len(thresholds)
    if products_plot:
        warnings.warn("run_validation: plotting (products_plot) is not part of the HIP build; skipped")
    model.eval()
    device = model.device

    if path_save_results is not None and "://" in str(path_save_results):
        # validation.py:213-219 writes to a temporary folder and uploads it with fsspec.  Cloud-bucket I/O is outside this build
        # (DESIGN.md scope table): refuse BEFORE the run, not after it when the results would be lost with the exception
        raise NotImplementedError(f"run_validation: remote result folder {path_save_results!r} (validation.py:213-219) is outside "
                                  "the hot path; pass a local path_save_results and copy it yourself")

    # how the model turns a prediction into a mask at a threshold: a known (threshold, opening) spec runs all
    # thresholds in one kernel; an opaque apply_threshold is called per threshold like the reference does
    spec = model.threshold_spec() if hasattr(model, "threshold_spec") else (None if hasattr(model, "apply_threshold") else 0)
    cm_thr = torch.zeros((1, T, 2, 2), dtype=torch.int64, device=device)
    invalid = torch.zeros(1, dtype=torch.int64, device=device)
    tile_cm, scalars, ids = [], [], []
    for idx, plume_data in enumerate(dataloader):
        plume_data = model.batch_with_preds(to_device(plume_data, device))
        y = plume_data["output_norm"]
        assert y.shape[0] == 1, "This function is expected to run with batch_size 1"
        pb = plume_data["pred_binary"]
        ignore = None
        if mask_from_magic:
            assert "nodata_mask" in plume_data.keys()        # has to be provided by the dataloader
            ignore = plume_data["nodata_mask"][0]
        # per-tile matrix of the model's own mask (validation.py:84-106)
        tile_cm.append(threshold_confusion(pb.float(), y, [0.5], 0, ignore=ignore, invalid=invalid)[0, 0])
        # PR curve (validation.py:112-121): never masked
        if spec is None:
            for k, thr in enumerate(thresholds):
                mask_k = model.apply_threshold(plume_data["prediction"], thr)
                cm_thr[0, k] += threshold_confusion(mask_k.float(), y, [0.5], 0)[0, 0]
        else:
            threshold_confusion(plume_data["prediction"], y, thresholds, spec, out=cm_thr)
        y_long = y.long()
        scalars.append(torch.stack([y_long[0, 0].sum(), plume_data["has_plume"][0].long().reshape(()),
                                    plume_data["pred_classification"][0, 0].long(), pb[0, 0].sum()]))
        ids.append(plume_data["id"][0])

    if len(ids) == 0:
        raise ValueError("run_validation: empty dataloader")
    tile_cm = torch.stack(tile_cm).cpu()                       # the one read-back
    scalars = torch.stack(scalars).cpu()
    cm_thr = cm_thr[0].cpu()
    if int(invalid.item()) != 0:
        raise ValueError(f"run_validation: {int(invalid.item())} label pixels are not in {{0, 1}}")

    out_data = []
    funs = starcopmetrics.METRICS_CONFUSION_MATRIX + [starcopmetrics.TP, starcopmetrics.TN, starcopmetrics.FP, starcopmetrics.FN]
    for i, tile_id in enumerate(ids):
        row = {fun.__name__: fun(tile_cm[i]).item() for fun in funs}
        row["id"] = tile_id
        row["label_pixels_plume"] = scalars[i, 0].item()
        row["has_plume"] = scalars[i, 1].item()
        row["pred_classification"] = scalars[i, 2].item()
        row["pred_pixels_plume"] = scalars[i, 3].item()
        if verbose and products_plot:
            print(row)
        out_data.append(row)
    out_data = pd.DataFrame(out_data).set_index("id")

    # metrics by difficulty (validation.py:158-180): a tile "has a plume" iff its label has one; easy = more than 1000 px
    out_data["has_plume"] = out_data["label_pixels_plume"] > 0
    out_data["difficulty"] = out_data["label_pixels_plume"].apply(lambda x: "easy" if x > 1000 else "hard")
    by_diff = out_data.groupby(["has_plume", "difficulty"])[["TP", "FP", "TN", "FN"]].sum()
    by_diff["total"] = by_diff.sum(axis=1)
    by_diff["frac_total"] = by_diff["total"] / by_diff["total"].sum()

    metrics = {}
    item = by_diff.loc[(False, "hard")]
    metrics["FPR_no_plume"] = item.FP / (item.FP + item.TN)
    metrics["frac_total_easy"] = item.frac_total              # overwritten below, as in the reference (:169)
    for str_diff in ["easy", "hard"]:
        item = by_diff.loc[(True, str_diff)]
        cm_diff = torch.tensor([[item.TN, item.FP], [item.FN, item.TP]], requires_grad=False)
        for f in starcopmetrics.METRICS_CONFUSION_MATRIX:
            metrics[f"{f.__name__}_{str_diff}"] = f(cm_diff).item()
        metrics[f"frac_total_{str_diff}"] = item.frac_total

    cm = tile_cm.sum(dim=0)
    for fun in starcopmetrics.METRICS_CONFUSION_MATRIX:
        metrics[fun.__name__] = fun(cm).item()
    metrics["confusion_matrix"] = cm

    # tile classification (validation.py:190-199)
    cls_cm = starcopmetrics.BinaryConfusionMatrix()
    cls_cm.update(torch.from_numpy(out_data["pred_classification"].values).long(),
                  torch.from_numpy(out_data["has_plume"].values).long())
    cm_classification = cls_cm.compute()
    for fun in starcopmetrics.METRICS_CONFUSION_MATRIX:
        metrics[f"classification_{fun.__name__}"] = fun(cm_classification).item()
    metrics["classification_confusion_matrix"] = cm_classification

    metrics["thresholded"] = []
    for k, thr in enumerate(thresholds):
        d = {"threshold": thr, "confusion_matrix": cm_thr[k]}
        for fun in [starcopmetrics.precision, starcopmetrics.recall, starcopmetrics.TPR, starcopmetrics.FPR]:
            d[fun.__name__] = fun(cm_thr[k])
        metrics["thresholded"].append(d)

    if path_save_results is not None:
        os.makedirs(path_save_results, exist_ok=True)
        out_data.to_csv(os.path.join(path_save_results, "results.csv"))
        with open(os.path.join(path_save_results, "results_agg.json"), "w") as fh:
            json.dump(metrics, fh, cls=CustomJSONEncoder)
    return out_data, metrics


class CustomJSONEncoder(json.JSONEncoder):
    """pandas / numpy / torch values -> JSON types (validation.py:226-257)."""

    def default(self, o):
        if hasattr(o, "to_json"):
            return o.to_json()
        if isinstance(o, np.generic):
            return o.item()
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, pd.Timestamp):
            return o.isoformat()
        if hasattr(o, "numpy"):
            return o.numpy().tolist()
        return super().default(o)
