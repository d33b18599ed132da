"""End-to-end scene inference on the GPU: radiance cube -> mag1c -> network input -> plume mask.

The reference spreads this over its drivers and notebooks; the steps and their order are
  * EMIT (BASELINE configs[4]): ``mag1c_emit`` (starcop/models/mag1c_emit.py:16-90: bands in [2122, 2488] nm, float64
    filter on blocks of ``column_step`` columns) -> RGB = nearest bands to 640/550/460 nm -> range rescale of
    starcop/emit_tools/emit_dataset.py:62-106 -> ``model(x)`` -> sigmoid > 0.5;
  * AVIRIS-NG (configs[2]): ``run_mag1c`` (starcop/process_aviris.py:189-219: per detector column, alpha = 0) and the
    pre-computed RGB products -> ``padded_predict``.
Everything stays on the device between the stages; with ``torch.distributed`` initialised the column blocks of the
matched filter and the scenes are independent work items (``column_range`` / ``parallel.sharded_map``): no collective on
the data path.
"""
import numpy as np
import torch

from . import mag1c
from .features import emit_to_aviris_input
from .model_module import masks_from_logits

EMIT_MAG1C_RANGE_NM = (2122.0, 2488.0)          # mag1c_emit.py:40-43
RGB_NM = (640.0, 550.0, 460.0)


def nearest_bands(wavelengths, targets=RGB_NM):
    w = np.asarray(wavelengths, dtype=np.float64)
    return [int(np.argmin(np.abs(w - t))) for t in targets]


@torch.no_grad()
def emit_scene_predict(model, raw, wavelengths, template, fill_value=-9999.0, column_step=2, num_iter=30,
                       covariance_lerp_alpha=1e-4, column_range=None):
    """``raw``: (rows, cols, S) float32 EMIT L1B radiance (device or host), ``wavelengths``: (S,) nm, ``template``: unit CH4
    absorption for the bands inside [2122, 2488] nm (``mag1c.generate_template_from_bands``; (K,) or (K, 2)).
    Returns a dict of device tensors: ``mf`` (rows, cols) ppm*m, ``albedo``, ``input`` (4, H', W') in the AVIRIS value range,
    ``prediction`` (H', W') plume probability and ``pred_binary`` (int64) -- H', W' = rows, cols cropped to multiples of 32
    (emit_dataset.py:80-93)."""
    raw = torch.as_tensor(raw)
    dev = raw.device if raw.is_cuda else model.device
    raw = raw.to(dev).float()
    w = np.asarray(wavelengths, dtype=np.float64)
    keep = np.nonzero((w >= EMIT_MAG1C_RANGE_NM[0]) & (w <= EMIT_MAG1C_RANGE_NM[1]))[0]
    assert keep.size and np.all(np.diff(keep) == 1), "the mag1c bands must be contiguous"
    t = np.asarray(template, dtype=np.float64)
    t = t[:, 1] if t.ndim == 2 else t
    if t.size != keep.size:
        raise ValueError(f"template has {t.size} bands, the cube has {keep.size} inside {EMIT_MAG1C_RANGE_NM} nm")
    sub = raw[..., int(keep[0]):int(keep[-1]) + 1].contiguous()
    mf, alb = mag1c.mag1c_columns(sub, t, fill_value, column_step=column_step, num_iter=num_iter,
                                  covariance_lerp_alpha=covariance_lerp_alpha, column_range=column_range)
    rgb = raw[..., nearest_bands(w)].permute(2, 0, 1).contiguous()
    x = emit_to_aviris_input(mf, rgb)
    was = model.training
    model.eval()
    try:
        masks = masks_from_logits(model(x[None]))
    finally:
        model.train(was)
    return {"mf": mf, "albedo": alb, "input": x, "prediction": masks["prediction"][0, 0], "pred_binary": masks["pred_binary"][0, 0]}
