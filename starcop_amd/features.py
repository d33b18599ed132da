"""On-the-fly input features of the hot path, on the GPU.

Mirrors /root/reference/starcop/data/feature_extration.py:
  weight_mag1c :32-35, no_outliers :37-40, ratio_2c_match_c_from_sums_outlier :42-56 (e.g. the registered product
  ``ratio_aviris_2350_2310_out`` :196), and the EMIT->AVIRIS value-range rescale of
  starcop/emit_tools/emit_dataset.py:62-106 (the same constants as notebook inference_on_raw_EMIT_nc_file cell 17).
Tensors live on the device; tiles are batched (B, H, W) so that one launch handles a batch of tiles.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream


def _as_tiles(t):
    t = torch.as_tensor(t)
    _lib.require_device(t)
    t = t.contiguous().float()
    lead = t.shape[:-2]
    B = 1
    for d in lead:
        B *= int(d)
    return t, B, int(t.shape[-2] * t.shape[-1])


def trimmed_sums(x, p=5):
    """np.sum(no_outliers(tile, p)) for every (H, W) tile of ``x`` -> float64 tensor of the leading shape."""
    lib = _lib.load()
    x, B, n = _as_tiles(x)
    sums = torch.empty(B, dtype=torch.float64, device=x.device)
    wb = lib.sc_trimmed_sum_workspace_bytes(B)
    work = torch.empty(wb, dtype=torch.uint8, device=x.device)
    check(lib.sc_trimmed_sums(ptr(x), B, n, float(p), ptr(sums), ptr(work), wb, stream()))
    return sums.reshape(x.shape[:-2])


def ratio_2c_match_c_from_sums_outlier(background_channel, signal, p=5, zero_value_out=-.6):
    """Varon-style two-band ratio: R = (c*signal - background)/(background + 1e-6) with c matching the 5-95 % trimmed
    sums of the two bands; pixels where both bands are < 1e-6 get ``zero_value_out``.  Same argument order as the
    reference (it is called as ``f(band_absorbing, band_reference)``)."""
    lib = _lib.load()
    bg, B, n = _as_tiles(background_channel)
    sg, B2, n2 = _as_tiles(signal)
    if (B, n) != (B2, n2):
        raise ValueError("background and signal tiles must have the same shape")
    s_bg, s_sg = trimmed_sums(bg, p).reshape(-1), trimmed_sums(sg, p).reshape(-1)
    out = torch.empty_like(sg)
    check(lib.sc_band_ratio(ptr(bg), ptr(sg), ptr(out), B, n, ptr(s_bg), ptr(s_sg), 0.0, float(zero_value_out), stream()))
    return out


def _clip_scale(x, div, lo, hi, mult, nan_to_num=False):
    lib = _lib.load()
    x = torch.as_tensor(x)
    _lib.require_device(x)
    x = x.contiguous().float()
    out = torch.empty_like(x)
    check(lib.sc_clip_scale(ptr(x), ptr(out), x.numel(), float(div), float(lo), float(hi), float(mult), int(nan_to_num), stream()))
    return out


def weight_mag1c(mag1c):
    """Loss weight of a pixel: clip(mag1c / 400, 0.1, 1)."""
    return _clip_scale(mag1c, 400.0, 0.1, 1.0, 1.0)


# constants of emit_dataset.py:62-69
MAGIC_DIV_BY, RGB_DIV_BY, MAGIC_MULT_BY, RGB_MULT_BY = 240., 20., 1750., 60.


def emit_to_aviris_input(mf, rgb):
    """(H, W) mag1c + (3, H, W) RGB radiance of an EMIT scene -> (4, H', W') network input in the AVIRIS value range:
    crop to multiples of 32, clip(mf/240, 0, 2)*1750, clip(rgb/20, 0, 2)*60, nan_to_num."""
    mf, rgb = torch.as_tensor(mf), torch.as_tensor(rgb)
    h, w = (mf.shape[-2] // 32) * 32, (mf.shape[-1] // 32) * 32
    out = torch.empty((4, h, w), dtype=torch.float32, device=mf.device)
    out[0] = _clip_scale(mf[:h, :w], MAGIC_DIV_BY, 0.0, 2.0, MAGIC_MULT_BY, nan_to_num=True)
    out[1:] = _clip_scale(rgb[:, :h, :w], RGB_DIV_BY, 0.0, 2.0, RGB_MULT_BY, nan_to_num=True)
    return out
