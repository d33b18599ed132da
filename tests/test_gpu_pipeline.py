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


def test_tiled_inference_equals_whole_scene(hip):
    """sliding-window inference with a halo >= the receptive field (320 px) reproduces the whole-scene forward -- on the
    whole scene, seams included, within the 1e-4 logit contract (eval-mode BatchNorm is affine; tiles sit on the 32-px stride grid;
    windows are clipped at the scene border so zero padding falls where the whole-scene forward has it).  A short halo is an
    approximation: its error is reported and must stay confined to the pixels whose receptive field was cut."""
    torch.manual_seed(3)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    H, W = 1024, 1152
    x = torch.cat([torch.randn(1, H, W, generator=g).abs() * 600, torch.rand(3, H, W, generator=g) * 100 + 5]).to(DEV)
    with torch.no_grad():
        whole = model(x[None])[0, 0]
    tiled = pipeline.tiled_logits(model, x, tile=512, halo=pipeline.RECEPTIVE_HALO)
    e = relerr(tiled, whole)
    print(f"tiled (512 + halo 320) vs whole-scene logits on {H}x{W}: rel err {e:.2e}")
    assert e < 1e-4
    # masks are the same pixels (away from logit ~ 0)
    far = whole.abs() > 1e-3
    assert torch.equal((tiled >= 0)[far], (whole >= 0)[far])
    short = pipeline.tiled_logits(model, x, tile=512, halo=64)
    d = (short - whole).abs() / whole.abs().max()
    inner = torch.ones_like(d, dtype=torch.bool)
    for s0 in (512,):
        inner[max(0, s0 - 384):s0 + 384, :] = False
        inner[:, max(0, s0 - 384):s0 + 384] = False
    inner[:, 1024 - 384:1024 + 384] = False
    print(f"halo 64: max rel err {float(d.max()):.2e} overall, {float(d[inner].max()):.2e} further than 384 px from any seam")
    assert float(d[inner].max()) < 1e-4


def test_emit_scene_predict_tiled_and_ratio(hip):
    """configs[4] in one call: EMIT cube -> mag1c -> rescale -> tile-sharded sliding-window U-Net -> masks, plus the on-the-fly
    band ratio (feature_extration.py:42-56) on the nearest 2350 / 2310 nm bands, against the whole-scene path and the oracle"""
    rng = np.random.default_rng(1)
    rows, cols = 200, 170
    wl = np.linspace(381.0, 2493.0, 285)
    keep = np.nonzero((wl >= 2122.0) & (wl <= 2488.0))[0]
    templ = -np.abs(rng.standard_normal(keep.size)) * 0.3 - 0.05
    raw = (rng.uniform(1, 6, size=285) * (1 + 0.05 * rng.standard_normal((rows, cols, 285)))).astype(np.float32)
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(DEV).eval()
    a = pipeline.emit_scene_predict(model, raw, wl, templ, column_step=2, ratio_bands=(2350.0, 2310.0))
    b = pipeline.emit_scene_predict(model, raw, wl, templ, column_step=2, tile=64, halo=320)
    assert torch.equal(a["mf"], b["mf"]) and a["input"].shape == (4, 192, 160)
    assert relerr(b["prediction"], a["prediction"]) < 1e-4
    ia, ir = pipeline.nearest_bands(wl, (2350.0, 2310.0))
    want = host_ref.band_ratio(raw[..., ia], raw[..., ir])
    assert relerr(a["ratio"], torch.as_tensor(want)) < 1e-5
    part = pipeline.emit_scene_predict(model, raw, wl, templ, column_step=2, column_range=(0, 80))
    assert set(part) == {"mf", "albedo"} and torch.equal(part["mf"][:, :80], a["mf"][:, :80])
    with pytest.raises(ValueError):
        pipeline.emit_scene_predict(model, raw, wl, templ, column_step=2, column_range=(1, 80))
