"""EMIT scene pipeline (mag1c_columns -> RGB bands -> emit range rescale -> network -> masks) against the same steps done by
hand with the CPU oracle pieces: every stage stays on the device and the composition equals the manual one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hip_ops import DEV, relerr  # noqa: E402
from oracle import host_ref, mag1c_ref  # noqa: E402
from starcop_amd import model_module as mm, pipeline  # noqa: E402


def test_emit_scene_predict(hip):
    rng = np.random.default_rng(0)
    rows, cols = 96, 70
    wl = np.linspace(381.0, 2493.0, 285)
    keep = np.nonzero((wl >= 2122.0) & (wl <= 2488.0))[0]
    S = keep.size
    templ = -np.abs(rng.standard_normal(S)) * 0.3 - 0.05
    base = rng.uniform(1, 6, size=285)
    raw = (base * (1 + 0.05 * rng.standard_normal((rows, cols, 285)))).astype(np.float32)
    raw[10:14, 3:5, :] = -9999.0                                   # fill pixels
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(DEV).eval()
    out = pipeline.emit_scene_predict(model, raw, wl, templ, column_step=2)
    # manual composition with the CPU restatements
    mf_ref, alb_ref = mag1c_ref.mag1c_columns(raw[..., keep[0]:keep[-1] + 1], templ, -9999.0, column_step=2)
    assert relerr(out["mf"], torch.as_tensor(mf_ref)) < 1e-4
    rgb = raw[..., pipeline.nearest_bands(wl)].transpose(2, 0, 1)
    x_ref = host_ref.emit_rescale(np.asarray(out["mf"].cpu()), rgb)
    assert out["input"].shape == (4, 96, 64) and relerr(out["input"], torch.as_tensor(x_ref)) < 1e-6
    with torch.no_grad():
        logits = model(out["input"][None])
    assert torch.equal(out["pred_binary"], (torch.sigmoid(logits[0, 0]) > 0.5).long())
    assert out["prediction"].shape == (96, 64) and float(out["prediction"].min()) >= 0.0
