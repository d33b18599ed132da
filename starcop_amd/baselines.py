"""Thresholded-band baselines with morphological opening, on the HIP mask kernels.

Mirrors /root/reference/starcop/baselines.py: ``binary_opening`` :25-27, ``Mag1cBaseline`` :31-77,
``SanchezBaseline`` :81-138, ``VaronBaseline`` :141-197 (same constructor arguments, ``forward``, ``apply_threshold``,
``batch_with_preds`` keys).  The reference builds the opening from kornia's erosion/dilation (two unfold passes and a
float round-trip per threshold); here it is one fused kernel over the thresholded band (``sc_binary_opening``) and
``run_validation`` sweeps all PR thresholds in a single pass (``sc_threshold_confusion``).
"""
from typing import Dict, List

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .model_module import Settings, differences
from .normalizer import DataNormalizer

ELEMENT_STRONGER = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.float32)     # baselines.py:39-41


def se_bits(kernel) -> int:
    """3x3 structuring element -> the ``se_bits`` of the C ABI (bit 3*r + c set <=> kernel[r][c] != 0)."""
    k = kernel.detach().cpu().numpy() if torch.is_tensor(kernel) else np.asarray(kernel)
    if k.shape != (3, 3):
        raise ValueError(f"only 3x3 structuring elements are supported, got {k.shape}")
    bits = 0
    for r in range(3):
        for c in range(3):
            if k[r, c] != 0:
                bits |= 1 << (3 * r + c)
    if bits == 0:
        raise ValueError("empty structuring element")
    return bits


def thresholded_opening(pred: torch.Tensor, threshold: float, bits: int, with_count=False):
    """int64 mask ``opening(pred > threshold)`` of a (..., H, W) float tensor (``bits`` = 0: no morphology)."""
    _lib.require_device(pred)
    lib = _lib.load()
    x = pred.contiguous().float()
    H, W = x.shape[-2:]
    n = x.numel() // (H * W)
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    cnt = torch.zeros(n, dtype=torch.int64, device=x.device) if with_count else None
    check(lib.sc_binary_opening(ptr(x), float(np.float32(threshold)), int(bits), ptr(out), ptr(cnt), n, H, W, stream()))
    return (out, cnt) if with_count else out


def binary_opening(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """bool mask: dilation(erosion(x)) of a binary (B,C,H,W) tensor by ``kernel`` (baselines.py:25-27)."""
    return thresholded_opening(x.float(), 0.0, se_bits(kernel)) > 0


class _BandThresholdBaseline(torch.nn.Module):
    """Common part of the three baselines: pick one input band, threshold it, optionally open the mask."""

    def __init__(self, input_products: List[str], band_name: str, threshold: float, use_normalisation: bool,
                 use_morphological_ops: bool):
        super().__init__()
        self.band_baseline = input_products.index(band_name)
        self.baseline_threshold = threshold
        self.element_stronger = torch.nn.Parameter(torch.from_numpy(ELEMENT_STRONGER.copy()), requires_grad=False)
        self.normalizer = DataNormalizer(Settings(dataset=dict(input_products=list(input_products),
                                                               output_products=["labelbinary"])))
        self.use_normalisation = use_normalisation
        self.use_morphological_ops = use_morphological_ops

    @property
    def device(self):
        return self.element_stronger.device

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x[:, self.band_baseline:(self.band_baseline + 1)]

    def threshold_spec(self) -> int:
        """``se_bits`` of ``apply_threshold`` (0 = plain ``pred > threshold``): lets ``run_validation`` sweep its PR
        thresholds in one kernel instead of one ``apply_threshold`` call per threshold."""
        return se_bits(self.element_stronger) if self.use_morphological_ops else 0

    def apply_threshold(self, pred: torch.Tensor, threshold) -> torch.Tensor:
        return thresholded_opening(pred, threshold, self.threshold_spec())

    def batch_with_preds(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        from .model_module import pred_classification
        batch = batch.copy()
        batch["input_norm"] = self.normalizer.normalize_x(batch["input"])
        batch["output_norm"] = self.normalizer.normalize_y(batch["output"])
        pred = self(batch["input_norm"] if self.use_normalisation else batch["input"])
        batch["prediction"] = pred
        batch["pred_binary"] = self.apply_threshold(pred, self.baseline_threshold)
        batch["differences"] = differences(batch["pred_binary"], batch["output_norm"].long())
        batch["pred_classification"] = pred_classification(batch["pred_binary"])
        return batch


class Mag1cBaseline(_BandThresholdBaseline):
    """mag1c band > 500 ppm*m, opened with the 3x3 cross (baselines.py:31-77; the band is taken un-normalised)."""

    def __init__(self, input_products: List[str], mag1c_threshold: float = 500.0):
        super().__init__(input_products, "mag1c", mag1c_threshold, use_normalisation=False, use_morphological_ops=True)
        self.band_mag1c = self.band_baseline
        self.mag1c_threshold = mag1c_threshold


class SanchezBaseline(_BandThresholdBaseline):
    """WV3 B8-vs-MLR ratio baseline (baselines.py:81-138)."""

    def __init__(self, input_products: List[str], baseline_threshold: float = 0.05, use_normalisation=True,
                 use_morphological_ops=True, band_name="ratio_wv3_B8_B8MLR_SanchezGarcia22_sum_c_out"):
        super().__init__(input_products, band_name, baseline_threshold, use_normalisation, use_morphological_ops)


class VaronBaseline(_BandThresholdBaseline):
    """WV3 B7/B5 ratio baseline (baselines.py:141-197)."""

    def __init__(self, input_products: List[str], baseline_threshold: float = 0.05, use_normalisation=True,
                 use_morphological_ops=True):
        super().__init__(input_products, "ratio_wv3_B7_B5_varon21_sum_c_out", baseline_threshold, use_normalisation,
                         use_morphological_ops)
