"""Evaluation masks and run_validation on the GPU against the numpy oracle: integer masks and confusion
counts are bit-exact, the float metrics equal the oracle's float64 values to float32 rounding (the reference divides
int64 tensors, i.e. in float32)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import host_ref  # noqa: E402
from starcop_amd import baselines, validation  # noqa: E402
from starcop_amd._lib import SE_CROSS  # noqa: E402

DEV = "cuda"
CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
SES = [CROSS, np.ones((3, 3)), np.array([[1, 1, 0], [0, 1, 0], [0, 0, 0]]), np.array([[0, 0, 0], [0, 1, 1], [1, 0, 0]])]


def _blobs(rng, H, W, n=6, amp=1200.0):
    """mag1c-like field: noise + a few Gaussian plumes, so that thresholding gives blobs, specks and holes."""
    yy, xx = np.mgrid[0:H, 0:W]
    f = np.abs(rng.normal(0, 260, size=(H, W)))
    for _ in range(n):
        cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(1.5, max(2.0, min(H, W) / 6))
        f += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    return f.astype(np.float32)


def test_se_bits_cross():
    assert baselines.se_bits(CROSS) == SE_CROSS == 0xBA


@pytest.mark.parametrize("shape", [(1, 1), (3, 200), (37, 53), (64, 64), (130, 67), (512, 512)])
def test_binary_opening_matches_oracle(hip, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    H, W = shape
    x = np.stack([_blobs(rng, H, W), _blobs(rng, H, W, n=2)])
    x[0, 0, :] = 900.0                                            # a line on the image border
    xt = torch.from_numpy(x).to(DEV)
    for se in SES:
        for thr in (500.0, 300.0):
            got, cnt = baselines.thresholded_opening(xt, thr, baselines.se_bits(se), with_count=True)
            got = got.cpu().numpy()
            assert got.dtype == np.int64
            for b in range(2):
                want = host_ref.apply_threshold(x[b], thr, se)
                assert np.array_equal(got[b] != 0, want), (shape, se.tolist(), thr)
            assert np.array_equal(cnt.cpu().numpy(), got.reshape(2, -1).sum(1))
    plain = baselines.thresholded_opening(xt, 500.0, 0).cpu().numpy()
    assert np.array_equal(plain, (x > 500.0).astype(np.int64))
    # the reference's entry point: bool mask in, bool mask out
    m = torch.from_numpy(x > 400.0).to(DEV)[:, None]
    ob = baselines.binary_opening(m, torch.from_numpy(CROSS.astype(np.float32)))
    assert ob.dtype == torch.bool and ob.shape == m.shape
    assert np.array_equal(ob.cpu().numpy()[0, 0], host_ref.binary_opening(x[0] > 400.0, CROSS))


def test_threshold_confusion_all_thresholds_one_pass(hip):
    rng = np.random.default_rng(5)
    N, H, W = 3, 150, 97
    p = rng.uniform(0, 1, size=(N, 1, H, W)).astype(np.float32)
    p[0, 0, :4] = [[0.5] * W, [0.95] * W, [np.nan] * W, [0.999] * W]        # ties with thresholds: strict >, NaN never
    y = (rng.uniform(size=(N, 1, H, W)) < 0.3).astype(np.float32)
    ig = (rng.uniform(size=(N, 1, H, W)) < 0.1)
    thr = np.sort([0, 1e-3, 1e-2] + np.arange(0.5, .96, .05).tolist() + [.99, .995, .999])[::-1]
    for se, bits in ((None, 0), (CROSS, SE_CROSS)):
        for ignore in (None, ig):
            cm = validation.threshold_confusion(torch.from_numpy(p).to(DEV), torch.from_numpy(y).to(DEV), thr, bits,
                                                ignore=None if ignore is None else torch.from_numpy(ignore).to(DEV)).cpu().numpy()
            assert cm.shape == (N, len(thr), 2, 2)
            for n in range(N):
                for k, t in enumerate(thr):
                    want = host_ref.confusion(host_ref.apply_threshold(p[n, 0], t, se), y[n, 0],
                                              None if ignore is None else ignore[n, 0])
                    assert np.array_equal(cm[n, k], want), (bits, n, t)
    # more thresholds than one launch takes, accumulation into a caller buffer, and labels outside {0,1}
    many = np.linspace(0.01, 0.99, 45)
    out = torch.ones((N, 45, 2, 2), dtype=torch.int64, device=DEV)
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    y2 = y.copy(); y2[1, 0, 5, :7] = 2.0; y2[2, 0, 0, 0] = -1.0
    validation.threshold_confusion(torch.from_numpy(p).to(DEV), torch.from_numpy(y2).to(DEV), many, 0, out=out, invalid=bad)
    assert int(bad.item()) == 8
    ign2 = (y2 != 0) & (y2 != 1)
    for n in range(N):
        for k in (0, 31, 32, 44):
            want = host_ref.confusion(p[n, 0] > np.float32(many[k]), np.where(ign2[n, 0], 0, y2[n, 0]), ign2[n, 0])
            assert np.array_equal(out[n, k].cpu().numpy() - 1, want)


def _tiles(rng, n_tiles, H=96, W=96):
    """tiles covering every (has_plume, difficulty) group of run_validation"""
    batches = []
    for i in range(n_tiles):
        kind = i % 3                                   # 0: no plume, 1: large plume (> 1000 px), 2: small plume
        mag = np.abs(rng.normal(0, 200, size=(H, W))).astype(np.float32)
        lab = np.zeros((H, W), np.float32)
        if kind:
            r = 30 if kind == 1 else 6
            cy, cx = rng.integers(r, H - r), rng.integers(r, W - r)
            yy, xx = np.mgrid[0:H, 0:W]
            blob = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
            lab[blob] = 1.0
            mag += 900.0 * np.roll(blob, (rng.integers(-3, 4), rng.integers(-3, 4)), (0, 1))
        if i == 0:
            mag[10:20, 10:25] += 800.0                 # false positives on a plume-free tile
        rgb = rng.uniform(5, 110, size=(3, H, W)).astype(np.float32)
        batches.append({"input": torch.from_numpy(np.concatenate([mag[None], rgb])[None]),
                        "output": torch.from_numpy(lab[None, None]), "id": [f"tile_{i:02d}"],
                        "has_plume": torch.tensor([int(kind != 0)]),
                        "weight_loss": torch.from_numpy(host_ref.weight_mag1c(mag)[None, None]),
                        "nodata_mask": torch.from_numpy((rng.uniform(size=(1, 1, H, W)) < 0.05).astype(np.uint8))})
    return batches


PRODUCTS = ["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"]


def _check_against_oracle(df, met, rows, want, masked=False):
    assert list(df.index) == [f"tile_{i:02d}" for i in range(len(rows))]
    for i, r in enumerate(rows):
        got = df.iloc[i]
        for k in ("TP", "TN", "FP", "FN", "label_pixels_plume", "pred_pixels_plume", "pred_classification"):
            assert int(got[k]) == int(r[k]), (i, k)
        assert bool(got["has_plume"]) == r["has_plume"] and got["difficulty"] == r["difficulty"]
        for k in ("precision", "recall", "f1score", "iou", "accuracy", "cohen_kappa", "balanced_accuracy"):
            a, b = float(got[k]), float(r[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2e-6 * max(1.0, abs(b)), (i, k, a, b)
    assert np.array_equal(met["confusion_matrix"].numpy(), want["confusion_matrix"])
    assert np.array_equal(met["classification_confusion_matrix"].numpy(), want["classification_confusion_matrix"])
    keys = [k for k in met if k not in ("confusion_matrix", "classification_confusion_matrix", "thresholded")]
    assert {"FPR_no_plume", "f1score", "iou_easy", "recall_hard", "frac_total_easy", "frac_total_hard",
            "classification_f1score", "cohen_kappa"} <= set(keys)
    for k in keys:
        a, b = float(met[k]), float(want[k])
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2e-6 * max(1.0, abs(b)), (k, a, b)
    assert len(met["thresholded"]) == len(want["thresholded"])
    for g, w in zip(met["thresholded"], want["thresholded"]):
        assert float(g["threshold"]) == w["threshold"]
        assert np.array_equal(g["confusion_matrix"].numpy(), w["confusion_matrix"])
        m = host_ref.metrics(w["confusion_matrix"])
        for k, kk in (("precision", "precision"), ("recall", "recall"), ("TPR", "recall"), ("FPR", "FPR")):
            a, b = float(g[k]), float(m[kk])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2e-6


@pytest.mark.parametrize("masked", [False, True])
def test_run_validation_mag1c_baseline(hip, tmp_path, masked):
    rng = np.random.default_rng(11)
    batches = _tiles(rng, 7)
    model = baselines.Mag1cBaseline(PRODUCTS).to(DEV)
    thr = [200.0, 350.0, 500.0, 700.0, 1000.0]
    with np.errstate(all="ignore"):
        df, met = validation.run_validation(model, batches, thresholds=thr, path_save_results=str(tmp_path),
                                            mask_from_magic=masked)
        preds = [b["input"][0, 0].numpy() for b in batches]
        pbs = [host_ref.apply_threshold(p, 500.0, CROSS) for p in preds]
        labels = [b["output"][0, 0].numpy() for b in batches]
        rows, want = host_ref.run_validation(preds, pbs, labels, thr, se=CROSS,
                                             ignores=[b["nodata_mask"][0, 0].numpy() for b in batches] if masked else None)
    _check_against_oracle(df, met, rows, want)
    # batch_with_preds keys and values of the baseline (baselines.py:59-75)
    out = model.batch_with_preds(validation.to_device(batches[1], DEV))
    assert {"input_norm", "output_norm", "prediction", "pred_binary", "differences", "pred_classification"} <= set(out)
    assert np.array_equal(out["pred_binary"][0, 0].cpu().numpy(), pbs[1].astype(np.int64))
    assert np.array_equal(out["differences"][0, 0].cpu().numpy(), host_ref.differences(pbs[1].astype(np.int64), labels[1]))
    assert out["pred_classification"].shape == (1, 1) and int(out["pred_classification"]) == rows[1]["pred_classification"]
    assert torch.equal(out["prediction"], out["input"][:, 0:1])
    # files of validation.py:213-216
    saved = json.load(open(os.path.join(tmp_path, "results_agg.json")))
    assert saved["confusion_matrix"] == want["confusion_matrix"].tolist()
    assert len(saved["thresholded"]) == 5 and saved["thresholded"][0]["threshold"] == 1000.0
    import pandas as pd
    csv = pd.read_csv(os.path.join(tmp_path, "results.csv")).set_index("id")
    assert list(csv.index) == list(df.index) and int(csv.loc["tile_01", "TP"]) == rows[1]["TP"]


def test_run_validation_unet_model_and_opaque_apply_threshold(hip):
    """ModelModule (plain thresholds in one pass) and a model with an opaque apply_threshold (called per threshold)."""
    from starcop_amd.model_module import ModelModule, default_settings
    torch.manual_seed(0)
    rng = np.random.default_rng(2)
    batches = _tiles(rng, 6, 64, 64)
    model = ModelModule(default_settings()).to(DEV)
    with np.errstate(all="ignore"):
        df, met = validation.run_validation(model, batches)
        model.eval()
        preds, pbs = [], []
        for b in batches:
            o = model.batch_with_preds(validation.to_device(b, DEV))
            preds.append(o["prediction"][0, 0].cpu().numpy()); pbs.append(o["pred_binary"][0, 0].cpu().numpy())
        rows, want = host_ref.run_validation(preds, pbs, [b["output"][0, 0].numpy() for b in batches])
    _check_against_oracle(df, met, rows, want)
    assert len(met["thresholded"]) == 16 and float(met["thresholded"][0]["threshold"]) == 0.999

    class Opaque(torch.nn.Module):
        """exposes apply_threshold only (no threshold_spec): run_validation must call it once per threshold"""

        def __init__(self, inner):
            super().__init__()
            self.inner, self.calls = inner, 0
            self.device = inner.device
            self.batch_with_preds = inner.batch_with_preds

        def apply_threshold(self, pred, threshold):
            self.calls += 1
            return self.inner.apply_threshold(pred, threshold)
    op = Opaque(baselines.Mag1cBaseline(PRODUCTS).to(DEV))
    assert not hasattr(op, "threshold_spec")
    batches = _tiles(np.random.default_rng(11), 7)
    with np.errstate(all="ignore"):
        df2, met2 = validation.run_validation(op, batches, thresholds=[300.0, 500.0])
        df1, met1 = validation.run_validation(baselines.Mag1cBaseline(PRODUCTS).to(DEV), batches, thresholds=[300.0, 500.0])
    for a, b in zip(met1["thresholded"], met2["thresholded"]):
        assert torch.equal(a["confusion_matrix"], b["confusion_matrix"])
    assert df1.equals(df2) and op.calls == 2 * 7
