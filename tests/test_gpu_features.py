"""Band-ratio / weight / EMIT-rescale kernels against the reference's own outputs (tests/golden/g8_ratio.npz) and the
numpy oracle on seeded tiles (tolerance 1e-5 absolute on the ratio: the reference sums in float32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import host_ref  # noqa: E402
from starcop_amd import features  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def test_band_ratio_vs_reference_golden(hip):
    g = np.load(os.path.join(G, "g8_ratio.npz"))
    out = features.ratio_2c_match_c_from_sums_outlier(torch.from_numpy(g["bg"]).to(DEV), torch.from_numpy(g["sig"]).to(DEV)).cpu().numpy()
    assert out.shape == g["ratio"].shape and out.dtype == np.float32
    assert np.abs(out - g["ratio"]).max() < 1e-5
    assert (out[0, :4, :4] == np.float32(-0.6)).all()                 # zero/zero pixels
    w = features.weight_mag1c(torch.from_numpy(g["mag1c"]).to(DEV)).cpu().numpy()
    assert np.array_equal(w, g["weight"])


def test_trimmed_sums_and_batched_tiles(hip):
    rng = np.random.default_rng(3)
    x = rng.uniform(0.2, 4.0, size=(5, 512, 512)).astype(np.float32)
    x[1, :10] = 0.0                      # ties at the lower percentile
    x[2, 0, :7] = [1e4, -3.0, 5e3, -1.0, 9e3, 0.0, 2e4]
    got = features.trimmed_sums(torch.from_numpy(x).to(DEV)).cpu().numpy()
    want = np.array([host_ref.trimmed(t.ravel(), 5).astype(np.float64).sum() for t in x])
    assert np.abs(got - want).max() / want.max() < 1e-9
    bg = (x * rng.uniform(0.9, 1.2, size=x.shape)).astype(np.float32)
    R = features.ratio_2c_match_c_from_sums_outlier(torch.from_numpy(bg).to(DEV), torch.from_numpy(x).to(DEV)).cpu().numpy()
    for b in range(5):
        assert np.abs(R[b] - host_ref.band_ratio(bg[b].copy(), x[b].copy())).max() < 2e-5


def test_emit_rescale(hip):
    rng = np.random.default_rng(4)
    mf = rng.uniform(-50, 900, size=(70, 100)).astype(np.float32)
    rgb = rng.uniform(-1, 60, size=(3, 70, 100)).astype(np.float32)
    mf[3, 3] = np.nan; rgb[1, 5, 5] = np.inf
    got = features.emit_to_aviris_input(torch.from_numpy(mf).to(DEV), torch.from_numpy(rgb).to(DEV)).cpu().numpy()
    want = host_ref.emit_rescale(mf, rgb)
    assert got.shape == (4, 64, 96)
    assert np.abs(got - want).max() < 1e-3 and np.isfinite(got).all()


def test_trimmed_sums_scene_sized_tile(hip):
    """a whole EMIT scene as ONE tile (1280 x 1242 = 1.6 M pixels): the multi-work-group radix select (global histograms per pass)
    gives the same exact order statistics as numpy; just below its size threshold the single-work-group select runs -- same answer"""
    rng = np.random.default_rng(9)
    for shape in ((1, 1280, 1242), (2, 361, 363), (3, 300, 436)):          # 1.6 M (large path), 131 043 (small path), 130 800
        x = rng.normal(1.0, 2.0, size=shape).astype(np.float32)
        x[0, :5] = -7.5                                                     # ties
        x[0, 7, :9] = [3e4, -2e4, 0.0, 1e-30, -1e-30, 5.0, 5.0, 5.0, 1e9]
        got = features.trimmed_sums(torch.from_numpy(x).to(DEV)).cpu().numpy()
        want = np.array([host_ref.trimmed(t.ravel(), 5).astype(np.float64).sum() for t in x])
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-9, shape
