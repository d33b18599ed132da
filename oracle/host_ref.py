"""CPU restatement of the small host-side pieces of the hot path (numpy).

TEST INFRASTRUCTURE -- not product code.  PINNED against the reference's own outputs in tests/golden/
(g4_normalizer, g5_masks, g6_padding, g7_metrics, g8_ratio; written by tests/golden/make_golden.py).

  normalize_x / normalize_y   starcop/data/normalizer_module.py:7-74 (table), :134-144
  pred_classification          starcop/models/model_module.py:210-212
  differences                  starcop/models/model_module.py:268-269
  find_padding / padded_predict starcop/models/utils/padding.py:5-50
  confusion-matrix metrics     starcop/metrics.py:20-85
  band ratio / weight_mag1c    starcop/data/feature_extration.py:32-56
  emit_rescale                 starcop/emit_tools/emit_dataset.py:62-106
"""
import numpy as np

NORM = {"mag1c": (0, 1750, 0, 2)}
for _b in ("550nm", "640nm", "460nm"):
    NORM[f"TOA_AVIRIS_{_b}"] = (0, 60, 0, 2)
NORM.update({"TOA_AVIRIS_2004nm": (0, 1, 0, 2), "TOA_AVIRIS_2109nm": (0, 5, 0, 2), "TOA_AVIRIS_2310nm": (0, 4, 0, 2),
             "TOA_AVIRIS_2350nm": (0, 3, 0, 2), "TOA_AVIRIS_2360nm": (0, 3, 0, 2),
             "ratio_aviris_2350_2310_out": (0, 0.0625, -2., 2.), "ratio_aviris_2350_2360_out": (0, 0.0625, -2., 2.),
             "ratio_aviris_2360_2310_out": (0, 0.0625, -2., 2.),
             "ratio_wv3_B8_B8MLR_SanchezGarcia22_sum_c_out": (0, 0.0769, -2., 2.),
             "ratio_wv3_B8_B8MLR_SanchezGarcia22_simplediv": (-0.5, 1, -2., 2.)})


def normalize_x(x, products):
    """(B,C,H,W) float32 -> float32; per-channel (x - offset) / factor clipped; unknown product: factor 1, clip +-10.
    Integer table entries keep the arithmetic in float32, float entries promote to float64 first (torch type promotion
    with the 0-dim-less parameter tensors of the reference), then the result is cast to float32."""
    rows = [NORM.get(p, (0, 1, -10, 10)) for p in products]
    as_int = all(isinstance(v, (int, np.integer)) for r in rows for v in r)
    pdt = np.int64 if as_int else np.float64
    off, fac, lo, hi = (np.array([r[i] for r in rows], dtype=pdt)[None, :, None, None] for i in range(4))
    xx = x if as_int else x.astype(np.float64)
    v = (xx - off) / fac if not as_int else (x - off.astype(np.float32)) / fac.astype(np.float32)
    return np.clip(v, lo, hi).astype(np.float32)


def pred_classification(pred_binary):
    h, w = pred_binary.shape[-2:]
    return (pred_binary.sum(axis=(-1, -2)) > (10 * h * w) / 64 ** 2).astype(np.int64)


def differences(pred_binary, gt):
    return 2 * pred_binary.astype(np.int64) + (gt == 1).astype(np.int64)


def find_padding(v, divisor=8):
    tgt = max(divisor, int(divisor * np.ceil(v / divisor)))
    a = (tgt - v) // 2
    return a, tgt - v - a


def padded_predict(x, model, divisor=32):
    (t, b), (l, r) = find_padding(x.shape[-2], divisor), find_padding(x.shape[-1], divisor)
    out = model(np.pad(x, ((0, 0), (t, b), (l, r)), "reflect")[None])[0]
    return out[..., t:t + x.shape[-2], l:l + x.shape[-1]]


def metrics(cm):
    cm = np.asarray(cm, dtype=np.float64)
    tn, fp, fn, tp = cm[0, 0], cm[0, 1], cm[1, 0], cm[1, 1]
    prec, rec = tp / (tp + fp), tp / (tp + fn)
    tot = cm.sum()
    exp_off = (cm.sum(1)[0] * cm.sum(0)[1] + cm.sum(1)[1] * cm.sum(0)[0]) / tot
    return {"precision": prec, "recall": rec, "f1score": 2 * prec * rec / (prec + rec), "iou": tp / (tp + fn + fp),
            "accuracy": (tp + tn) / tot, "cohen_kappa": 1 - (fp + fn) / exp_off,
            "balanced_accuracy": 0.5 * (rec + tn / (tn + fp)), "TP": tp, "TN": tn, "FP": fp, "FN": fn, "FPR": fp / (fp + tn)}


def trimmed(d, p=5):
    lo, hi = np.percentile(d, p), np.percentile(d, 100 - p)
    return d[(d >= lo) & (d <= hi)]


def band_ratio(background, signal, p=5, zero_value_out=-0.6):
    """(c*signal - background)/(background + 1e-6), c = trimmed-sum(background)/trimmed-sum(signal); 0/0 -> -0.6."""
    c = trimmed(background.ravel(), p).sum() / trimmed(signal.ravel(), p).sum()
    R = (c * signal - background) / (background + 1e-6)
    R[(signal < 1e-6) & (background < 1e-6)] = zero_value_out
    return R


def weight_mag1c(m):
    return np.clip(m / 400, 0.1, 1)


def emit_rescale(mf, rgb):
    """EMIT -> AVIRIS value range: crop to multiples of 32, clip(mf/240,0,2)*1750, clip(rgb/20,0,2)*60, nan_to_num."""
    h, w = (mf.shape[0] // 32) * 32, (mf.shape[1] // 32) * 32
    out = np.ones((4, h, w), dtype=np.float32)
    out[0] = np.clip(mf[:h, :w] / 240., 0., 2.) * 1750.
    out[1:] = np.clip(rgb[:, :h, :w] / 20., 0., 2.) * 60.
    return np.nan_to_num(out)


# ---- evaluation masks / run_validation -----------------------------------------------------------
# starcop/baselines.py:25-57 (binary_opening, Mag1cBaseline.apply_threshold) and starcop/validation.py:26-224.
# kornia (the reference's morphology backend) and torchmetrics are absent from this image and validation.py does not
# import without them: PARITY UNPINNED against the reference itself for this block; the opening is cross-checked against
# scipy.ndimage (tests/test_oracle.py) and the metric formulas are the golden-pinned ones above (g7_metrics).
def binary_opening(mask, se):
    """dilation(erosion(mask)) for a (H,W) boolean mask and a 3x3 structuring element, kornia 'geodesic' borders:
    outside pixels are +inf for the erosion and -inf for the dilation; the dilation uses the SE flipped in both axes."""
    mask = np.asarray(mask, dtype=bool)
    se = np.asarray(se) != 0
    H, W = mask.shape
    pad = np.ones((H + 2, W + 2), dtype=bool)
    pad[1:-1, 1:-1] = mask
    er = np.ones((H, W), dtype=bool)
    for r in range(3):
        for c in range(3):
            if se[r, c]:
                er &= pad[r:r + H, c:c + W]
    pad = np.zeros((H + 2, W + 2), dtype=bool)
    pad[1:-1, 1:-1] = er
    di = np.zeros((H, W), dtype=bool)
    sef = se[::-1, ::-1]
    for r in range(3):
        for c in range(3):
            if sef[r, c]:
                di |= pad[r:r + H, c:c + W]
    return di


def confusion(pred_binary, target, ignore=None):
    """[[TN, FP], [FN, TP]] int64, cm[target, prediction]."""
    p = np.asarray(pred_binary).astype(np.int64).reshape(-1)
    t = np.asarray(target).astype(np.int64).reshape(-1)
    if ignore is not None:
        keep = np.asarray(ignore).reshape(-1) == 0
        p, t = p[keep], t[keep]
    return np.bincount(t * 2 + p, minlength=4).reshape(2, 2)


def apply_threshold(pred, thr, se=None):
    m = np.asarray(pred, dtype=np.float32) > np.float32(thr)
    return binary_opening(m, se) if se is not None else m


def run_validation(preds, pred_binaries, labels, thresholds=None, se=None, ignores=None):
    """Per-tile rows and aggregated metrics of validation.py:26-224 from per-tile arrays: ``preds`` float (H,W) scores,
    ``pred_binaries`` the model's own masks, ``labels`` {0,1}.  Returns (rows: list of dict, metrics: dict)."""
    if thresholds is None:
        thresholds = [0, 1e-3, 1e-2] + np.arange(0.5, .96, .05).tolist() + [.99, .995, .999]
    thresholds = np.sort(thresholds)[::-1]
    rows, agg = [], np.zeros((2, 2), np.int64)
    cm_thr = np.zeros((len(thresholds), 2, 2), np.int64)
    for i, (p, pb, y) in enumerate(zip(preds, pred_binaries, labels)):
        cm = confusion(pb, y, None if ignores is None else ignores[i])
        agg += cm
        row = {k: v for k, v in metrics(cm.astype(np.float64)).items()}
        row.update(TP=int(cm[1, 1]), TN=int(cm[0, 0]), FP=int(cm[0, 1]), FN=int(cm[1, 0]))
        npl = int(np.asarray(y).astype(np.int64).sum())
        H, W = np.asarray(pb).shape
        row.update(label_pixels_plume=npl, has_plume=npl > 0, difficulty="easy" if npl > 1000 else "hard",
                   pred_pixels_plume=int(np.asarray(pb).sum()),
                   pred_classification=int(np.asarray(pb).sum() > 10 * H * W / 64 ** 2))
        rows.append(row)
        for k, thr in enumerate(thresholds):
            cm_thr[k] += confusion(apply_threshold(p, thr, se), y)
    out = {"confusion_matrix": agg}
    out.update(metrics(agg.astype(np.float64)))

    def group(has, diff):
        sel = [r for r in rows if r["has_plume"] == has and r["difficulty"] == diff]
        return {k: sum(r[k] for r in sel) for k in ("TP", "FP", "TN", "FN")}
    total = sum(r["TP"] + r["FP"] + r["TN"] + r["FN"] for r in rows)
    g = group(False, "hard")
    out["FPR_no_plume"] = g["FP"] / (g["FP"] + g["TN"])
    for d in ("easy", "hard"):
        g = group(True, d)
        m = metrics(np.array([[g["TN"], g["FP"]], [g["FN"], g["TP"]]], np.float64))
        out.update({f"{k}_{d}": v for k, v in m.items()})
        out[f"frac_total_{d}"] = sum(g.values()) / total
    ccm = confusion([r["pred_classification"] for r in rows], [int(r["has_plume"]) for r in rows])
    out["classification_confusion_matrix"] = ccm
    out.update({f"classification_{k}": v for k, v in metrics(ccm.astype(np.float64)).items()})
    out["thresholded"] = [dict(threshold=float(t), confusion_matrix=cm_thr[k]) for k, t in enumerate(thresholds)]
    return rows, out


# ---- training-batch assembly (starcop/data/datamodule.py:128-134, dataset.py:99-102) --------------------
# kornia 0.6.7 is absent (PARITY UNPINNED against it); rotate() there is warp_affine -> F.affine_grid/F.grid_sample with
# align_corners=True and zeros padding, about the centre ((w-1)/2, (h-1)/2).  tests/test_oracle.py checks this restatement
# against torch.nn.functional.grid_sample itself.
def rotate_flip(crop, cos_t, sin_t, rotate, hflip, vflip, nearest=False):
    """(C,h,w) float32 crop -> vflip(hflip(rotate(crop))) with the inverse map of sc_gather_augment, float32 arithmetic."""
    crop = np.asarray(crop, np.float32)
    C_, h, w = crop.shape
    out = crop
    if rotate:
        cx, cy = np.float32(0.5 * (w - 1)), np.float32(0.5 * (h - 1))
        yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        cs, sn = np.float32(cos_t), np.float32(sin_t)
        dx, dy = xx - cx, yy - cy
        xs, ys = cx + cs * dx - sn * dy, cy + sn * dx + cs * dy
        pad = np.zeros((C_, h + 2, w + 2), np.float32)
        pad[:, 1:-1, 1:-1] = crop

        def at(y, x):
            ok = (y >= 0) & (y < h) & (x >= 0) & (x < w)
            return np.where(ok, pad[:, np.clip(y, -1, h) + 1, np.clip(x, -1, w) + 1], np.float32(0))
        if nearest:
            out = at(np.rint(ys).astype(np.int64), np.rint(xs).astype(np.int64))
        else:
            x0, y0 = np.floor(xs), np.floor(ys)
            ax, ay = xs - x0, ys - y0
            x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
            one = np.float32(1)
            out = (at(y0, x0) * (one - ax) * (one - ay) + at(y0, x0 + 1) * ax * (one - ay)
                   + at(y0 + 1, x0) * (one - ax) * ay + at(y0 + 1, x0 + 1) * ax * ay)
    if hflip:
        out = out[:, :, ::-1]
    if vflip:
        out = out[:, ::-1, :]
    return np.ascontiguousarray(out, dtype=np.float32)
