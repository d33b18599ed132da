"""Input/label normalisation of the STARCOP products, mirroring
/root/reference/starcop/data/normalizer_module.py (table :7-74, ``DataNormalizer`` :78-149).

``normalize_x`` is ``clamp((x - offset) / factor, clip_min, clip_max).float()``.  On the hot path it is
not a separate pass: ``consts()`` hands the per-channel constants to the stem convolution, which applies
them while it loads the tile (SC_SRC_NORM).  The standalone methods run the same HIP prologue through
``sc_apply_src`` so ``batch_with_preds`` can return ``input_norm``.
"""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib
from ._lib import SC_CST, SRC_NORM, check, make_src, ptr, stream


def _aviris(f):
    return {"offset": 0, "factor": f, "clip": (0, 2)}


_UNIT = {"offset": 0, "factor": 1, "clip": (0, 2)}
BAND_NORMALIZATION = {}
for _sat in ("S2A", "S2B"):
    for _b in ("B1", "B2", "B3", "B4", "B5", "B6", "B7", "B8", "B8A", "B9", "B10", "B11", "B12"):
        BAND_NORMALIZATION[f"TOA_{_sat}_{_b}"] = dict(_UNIT)
for _k in range(1, 9):
    BAND_NORMALIZATION[f"TOA_WV3_SWIR{_k}"] = dict(_UNIT)
BAND_NORMALIZATION.update({
    "TOA_AVIRIS_550nm": _aviris(60), "TOA_AVIRIS_640nm": _aviris(60), "TOA_AVIRIS_460nm": _aviris(60),
    "TOA_AVIRIS_2004nm": _aviris(1), "TOA_AVIRIS_2109nm": _aviris(5), "TOA_AVIRIS_2310nm": _aviris(4),
    "TOA_AVIRIS_2350nm": _aviris(3), "TOA_AVIRIS_2360nm": _aviris(3),
    "mag1c": {"offset": 0, "factor": 1750, "clip": (0, 2)},
})
_R = lambda off, fac: {"offset": off, "factor": fac, "clip": (-2., 2.)}   # noqa: E731
BAND_NORMALIZATION.update({
    "ratio_aviris_2350_2310_out": _R(0, 0.0625), "ratio_aviris_2350_2360_out": _R(0, 0.0625),
    "ratio_aviris_2360_2310_out": _R(0, 0.0625),
    "ratio_wv3_B7_B5_varon21_sum_c_out": _R(0, 0.04), "ratio_wv3_B8_B5_varon21_sum_c_out": _R(0, 0.1),
    "ratio_wv3_B7_B6_varon21_sum_c_out": _R(0, 0.1),
    "ratio_wv3_B7_B7MLR_SanchezGarcia22_sum_c_out": _R(0, 0.025),
    "ratio_wv3_B8_B8MLR_SanchezGarcia22_sum_c_out": _R(0, 0.0769),
    "ratio_wv3_B7_B7MLR_SanchezGarcia22_simplediv": _R(0, 1),
    "ratio_wv3_B8_B8MLR_SanchezGarcia22_simplediv": _R(-0.5, 1),
    "ratio_lrn_bands2band8only_60ep_512_l1": _R(0, 0.5),
    "ratio_wv3_B7_B7MLR_fromS2_9bands_sum_c_out": _R(0, 1),
    "ratio_wv3_B7_B7MLR_fromS2_5bands_sum_c_out": _R(0, 0.1111111),
    "ratio_wv3_B8_B8MLR_fromS2_9bands_sum_c_out": _R(0, 0.125),
    "ratio_wv3_B8_B8MLR_fromS2_5bands_sum_c_out": _R(0, 0.1666666),
})


def _param(values):
    return torch.nn.Parameter(torch.from_numpy(np.array(values)[:, None, None]), requires_grad=False)


class DataNormalizer(torch.nn.Module):
    """Same constructor, parameters (names, dtypes, shapes) and methods as the reference class."""

    def __init__(self, settings):
        super().__init__()
        self.settings_dataset = settings.dataset
        off, fac, lo, hi = [], [], [], []
        for p in self.settings_dataset.input_products:
            if p not in BAND_NORMALIZATION:
                warnings.warn(f"Feature {p} does not have band normalization attributes. "
                              f"It will not be normalized BUT it will be clipped to [-10, 10]")
                off.append(0); fac.append(1); lo.append(-10); hi.append(10)
            else:
                e = BAND_NORMALIZATION[p]
                off.append(e["offset"]); fac.append(e["factor"]); lo.append(e["clip"][0]); hi.append(e["clip"][1])
        self.offsets_input, self.factors_input = _param(off), _param(fac)
        self.clip_min_input, self.clip_max_input = _param(lo), _param(hi)

        off, fac, lo, hi = [], [], [], []
        for p in self.settings_dataset.output_products:
            if p in BAND_NORMALIZATION:
                e = BAND_NORMALIZATION[p]
                off.append(e["offset"]); fac.append(e["factor"]); lo.append(e["clip"][0]); hi.append(e["clip"][1])
        if len(fac) > 0:
            assert len(fac) == len(self.settings_dataset.output_products), \
                "Some output products don't have normalization. CHECK!"
            self.factors_output, self.offsets_output = _param(fac), _param(off)
            self.clip_min_output, self.clip_max_output = _param(lo), _param(hi)
        else:
            self.factors_output = None
            self.offsets_output = None
        self._consts = None

    # ---- constants for the fused stem prologue: [C][8] = {offset, factor, clip_min, clip_max, 0...}
    def consts(self, device):
        c = self._consts
        # keyed on the parameters' identity and version: load_state_dict / in-place edits of the table invalidate the cache
        key = tuple((t.data_ptr(), t._version) for t in (self.offsets_input, self.factors_input, self.clip_min_input, self.clip_max_input))
        if c is None or c.device != torch.device(device) or getattr(self, "_consts_key", None) != key:
            self._consts_key = key
            n = self.offsets_input.shape[0]
            c = torch.zeros((n, SC_CST), dtype=torch.float32)
            c[:, 0] = self.offsets_input.reshape(-1).float().cpu()
            c[:, 1] = self.factors_input.reshape(-1).float().cpu()
            c[:, 2] = self.clip_min_input.reshape(-1).float().cpu()
            c[:, 3] = self.clip_max_input.reshape(-1).float().cpu()
            c = c.to(device)
            self._consts = c
        return c

    def normalize_x(self, x):
        _lib.require_device(x)
        lib = _lib.load()
        x = x.contiguous().float()
        squeeze = x.dim() == 3
        xv = x[None] if squeeze else x
        N, Cn, H, W = xv.shape
        out = torch.empty_like(xv)
        s = make_src(xv, Cn, SRC_NORM, cst=self.consts(x.device))
        check(lib.sc_apply_src(C.byref(s), ptr(out), N, Cn, H * W, stream()))
        return out[0] if squeeze else out

    def denormalize_x(self, x):
        return (x * self.factors_input) + self.offsets_input

    def normalize_y(self, y):
        if self.factors_output is not None:
            return torch.clamp((y - self.offsets_output) / self.factors_output,
                               self.clip_min_output, self.clip_max_output)
        return y

    def denormalize_y(self, y):
        if self.factors_output is not None:
            return (y * self.factors_output) + self.offsets_output
        return y
