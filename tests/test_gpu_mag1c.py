"""mag1c parity: the HIP filters (one launch for all groups, fp64 statistics) against
  (a) the golden vectors the reference itself produced (tests/golden/g1_filters.npz, g2_groups_*.npz), and
  (b) the CPU oracle (oracle/mag1c_ref.py) evaluated in float64 on the same inputs.
Tolerances: fp64 data: 1e-6 relative to max(|ref|,1) on every pixel; fp32 data vs the fp64 oracle:
1e-5; fp32 data vs the reference's own fp32 result: 1e-3 on >= 99.8 % of pixels (that is the reference's fp32 noise --
the filter is iterative and discontinuous).  NODATA / skipped-group patterns are exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mag1c_ref  # noqa: E402
from starcop_amd import mag1c as hip_mag1c  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / max(float(np.abs(b).max()), 1.0)


def _kw(g, name):
    kw = {}
    for k in g.files:
        if k.startswith(name + "_kw_"):
            v = g[k]
            kw[k[len(name) + 4:]] = v if v.ndim else v.item()
    return kw


def test_filters_vs_reference_golden_and_fp64_oracle(hip):
    g = np.load(os.path.join(G, "g1_filters.npz"))
    names = sorted({k[:-2] for k in g.files if k.endswith("_x") and not k.startswith("singular")})
    worst64 = worst32 = 0.0
    for name in names:
        x, t = g[name + "_x"], g[name + "_t"]
        kw = _kw(g, name)
        fn_hip = hip_mag1c.rmf if name.startswith("rmf") else hip_mag1c.acrwl1mf
        fn_ref = mag1c_ref.rmf if name.startswith("rmf") else mag1c_ref.acrwl1mf
        kw_t = {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        mf, R = fn_hip(torch.from_numpy(x).to(DEV), torch.from_numpy(t.astype(np.float64)), **kw_t)
        mf, R = mf.cpu().numpy(), R.cpu().numpy()
        assert mf.shape == g[name + "_mf"].shape and mf.dtype == x.dtype
        o_mf, o_R = fn_ref(x.astype(np.float64), t.astype(np.float64), **kw)        # fp64 oracle on the same data
        if x.dtype == np.float64:
            e = rel(mf, g[name + "_mf"]).max()
            worst64 = max(worst64, e)
            assert e < 1e-6, (name, e)
            if g[name + "_R"].size > 1:
                assert rel(R, g[name + "_R"]).max() < 1e-9, name
        else:
            e = rel(mf, o_mf).max()
            worst32 = max(worst32, e)
            assert e < 1e-5, (name, e)                       # against exact arithmetic on the same fp32 radiances
            d = rel(mf, g[name + "_mf"])                     # against the reference's fp32 run
            assert np.mean(d > 1e-3) < 2e-3, (name, float(d.max()))
            if g[name + "_R"].size > 1:
                assert rel(R, g[name + "_R"]).max() < 1e-5, name
    print("worst fp64", worst64, "worst fp32-vs-fp64-oracle", worst32)


def test_singular_covariance_raises(hip):
    g = np.load(os.path.join(G, "g1_filters.npz"))
    with pytest.raises(torch.linalg.LinAlgError):
        hip_mag1c.acrwl1mf(torch.from_numpy(g["singular_x"]).to(DEV), torch.from_numpy(g["singular_t"]), num_iter=2, alpha=0.0)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_group_driver_vs_reference_golden(hip, tag):
    g = np.load(os.path.join(G, f"g2_groups_{tag}.npz"))
    cube, groups, templ = g["cube"], g["groups"], g["templ"]
    flt = hip_mag1c.Filter(torch.from_numpy(templ.astype(np.float64)), num_iter=30, alpha=1e-4)
    mf, alb = hip_mag1c.func_by_groups(flt, torch.from_numpy(cube).to(DEV), groups)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    nod = g["mf"] == hip_mag1c.NODATA
    assert np.array_equal(mf == hip_mag1c.NODATA, nod)            # exact NODATA / <=10-pixel skip pattern
    assert np.array_equal(alb == hip_mag1c.NODATA, g["albedo"] == hip_mag1c.NODATA)
    d = rel(mf[~nod], g["mf"][~nod])
    if tag == "f64":
        assert d.max() < 1e-6
    else:
        assert np.mean(d > 1e-3) < 2e-3
    assert rel(alb[~nod], g["albedo"][~nod]).max() < 1e-5
    # explicit validity mask + generic-callable path (group by group, as the reference loops)
    mf2, _ = hip_mag1c.func_by_groups(lambda xg: hip_mag1c.acrwl1mf(xg, flt.template, num_iter=30, alpha=1e-4),
                                      torch.from_numpy(cube.clip(0.1, None)).to(DEV), groups, mask=g["mask2"])
    mf2 = mf2.cpu().numpy()
    nod2 = g["mf2"] == hip_mag1c.NODATA
    assert np.array_equal(mf2 == hip_mag1c.NODATA, nod2)
    d2 = rel(mf2[~nod2], g["mf2"][~nod2])
    assert (d2.max() < 1e-6) if tag == "f64" else (np.mean(d2 > 1e-3) < 2e-3)


def test_column_driver_emit_semantics(hip):
    """mag1c_emit core: float32 raw, fill pixels excluded, column blocks filtered independently in float64."""
    rng = np.random.default_rng(4)
    t = np.load(os.path.join(G, "g3_templates.npz"))["emit_template_kept"][:, 1]
    S, rows, cols = t.size, 96, 10
    base = rng.uniform(1, 6, size=S)
    raw = (base * (1 + 0.05 * rng.standard_normal((rows, cols, S)))).astype(np.float32)
    k = np.zeros((rows, cols)); k[20:40, 3:6] = 3e-5
    raw = (raw * (1 + k[..., None] * t)).astype(np.float32)
    raw[:7, :3, :] = -9999.0
    raw[50, 7, 5] = -9999.0
    want_mf, want_alb = mag1c_ref.mag1c_columns(raw, t, -9999.0, column_step=2, num_iter=30, alpha=1e-4)
    mf, alb = hip_mag1c.mag1c_columns(torch.from_numpy(raw).to(DEV), t, -9999.0, column_step=2, num_iter=30)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    assert mf.dtype == np.float32 and mf.shape == (rows, cols)
    assert np.array_equal(mf == -9999.0, want_mf == -9999.0)
    ok = want_mf != -9999.0
    assert rel(mf[ok], want_mf[ok]).max() < 1e-5 and rel(alb[ok], want_alb[ok]).max() < 1e-6
    # sharding the column blocks over ranks is a pure partition (no exchange): two shards == the whole
    a, _ = hip_mag1c.mag1c_columns(torch.from_numpy(raw).to(DEV), t, -9999.0, column_step=2, column_range=(0, 6))
    b, _ = hip_mag1c.mag1c_columns(torch.from_numpy(raw).to(DEV), t, -9999.0, column_step=2, column_range=(6, 10))
    merged = torch.where(a != -9999.0, a, b).cpu().numpy()
    assert np.array_equal(merged, mf)


def test_template_generated_on_the_box_drives_the_filter(hip):
    """a13 end to end on the GPU box: generate_template_from_bands reads the CH4 look-up table that ships with the package
    (starcop_amd/data/ch4.lut, no reference checkout needed), reproduces the reference's template (golden G3) and bad-band mask,
    and the template it made drives the EMIT column driver to the oracle's result."""
    g3 = np.load(os.path.join(G, "g3_templates.npz"))
    assert np.array_equal(hip_mag1c.get_mask_bad_bands(g3["badband_wave"]), g3["badband_keep"])
    full = hip_mag1c.generate_template_from_bands(g3["emit_centers"], g3["emit_fwhm"])
    t = full[g3["emit_keep"]]
    assert np.abs(t - g3["emit_template_kept"]).max() < 1e-9 * np.abs(g3["emit_template_kept"]).max()
    ta = hip_mag1c.generate_template_from_bands(g3["aviris_centers"], g3["aviris_fwhm"])[g3["aviris_keep"]]
    assert np.abs(ta - g3["aviris_template_kept"]).max() < 1e-9 * np.abs(g3["aviris_template_kept"]).max()
    rng = np.random.default_rng(14)
    S, rows, cols = t.shape[0], 64, 8
    raw = (rng.uniform(1, 6, size=S) * (1 + 0.05 * rng.standard_normal((rows, cols, S)))).astype(np.float32)
    k = np.zeros((rows, cols)); k[10:30, 2:5] = 3e-5
    raw = (raw * (1 + k[..., None] * t[:, 1])).astype(np.float32)
    want_mf, _ = mag1c_ref.mag1c_columns(raw, t[:, 1], -9999.0, column_step=2, num_iter=30, alpha=1e-4)
    mf, _ = hip_mag1c.mag1c_columns(torch.from_numpy(raw).to(DEV), t[:, 1], -9999.0, column_step=2, num_iter=30)
    assert rel(mf.cpu().numpy(), want_mf).max() < 1e-5


def test_column_layout_equals_sorted_layout(hip):
    """column-structured groups take the sort-free device layout (sc_mag1c_layout_columns); it must give exactly the packed
    order of the general path (stable sort by group id): same pixels, same order, same kernel -> bit-identical results, incl.
    NODATA pixels, a user mask, multi-column groups, a group of <= 10 valid pixels (skipped) and an all-invalid column"""
    rng = np.random.default_rng(31)
    t = np.load(os.path.join(G, "g3_templates.npz"))["aviris_template_kept"][:, 1][:24]
    H, W, S = 80, 37, 24
    cube = (rng.uniform(1, 6, size=S) * (1 + 0.05 * rng.standard_normal((H, W, S)))).astype(np.float32)
    cube[5:9, 3, 2] = hip_mag1c.NODATA                      # invalid pixels inside a group
    cube[:, 11, :] = hip_mag1c.NODATA                       # a whole column invalid
    cube[: H - 7, 15, 0] = hip_mag1c.NODATA                 # a single-column group with 7 valid pixels: skipped (<= 10)
    ids = np.concatenate([np.arange(1, 21), np.repeat(np.arange(30, 36), 3)])[:W]      # single columns, then 3-column blocks
    groups = np.repeat(ids[None, :], H, 0)
    x = torch.from_numpy(cube).to(DEV)
    mask = (rng.random((H, W)) > 0.1) & np.all(cube > hip_mag1c.NODATA, axis=-1)
    for m in (None, mask):
        hip_mag1c.COLUMN_FAST_PATH = True
        a_mf, a_alb = hip_mag1c.acrwl1mf_by_groups(x, t, groups, mask=m)
        try:
            hip_mag1c.COLUMN_FAST_PATH = False
            b_mf, b_alb = hip_mag1c.acrwl1mf_by_groups(x, t, groups, mask=m)
        finally:
            hip_mag1c.COLUMN_FAST_PATH = True
        assert torch.equal(a_mf, b_mf) and torch.equal(a_alb, b_alb)
        assert bool((a_mf[:, 11] == hip_mag1c.NODATA).all()) and bool((a_mf[:, 15] == hip_mag1c.NODATA).all())
        assert bool((a_mf[:, 0] != hip_mag1c.NODATA).all()) if m is None else True
    # and against the oracle
    want_mf, _ = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t, num_iter=30, alpha=0.0), cube, groups)
    got = a_mf if m is None else hip_mag1c.acrwl1mf_by_groups(x, t, groups)[0]
    got = got.cpu().numpy()
    assert np.array_equal(got == hip_mag1c.NODATA, want_mf == hip_mag1c.NODATA)
    ok = want_mf != hip_mag1c.NODATA
    assert np.mean(rel(got[ok], want_mf[ok]) > 1e-3) < 2e-3


@pytest.mark.parametrize("S,alpha", [(77, 0.0), (125, 0.0), (125, 1e-4)])
def test_direct_tiles_equal_the_packed_path(hip, S, alpha):
    """column groups of at most 512 pixels with more than 64 bands (a 512-row tile per detector column: configs[2]): the filter kernel
    gathers its pixels from the cube and writes image order itself (sc_mag1c_args.cube: no pack / scatter passes) -- bit-identical to the
    packed path, with invalid pixels inside groups, a skipped group, a user mask, a band slice of a wider cube, both alpha branches;
    groups that could exceed 512 pixels keep the packed path"""
    rng = np.random.default_rng(500 + S)
    H, W, S_total = 256, 13, S + 9                          # (a group needs well more valid pixels than bands)
    t = -np.abs(rng.standard_normal(S)) * 0.3
    cube = (rng.uniform(1, 6, size=S_total) * (1 + 0.05 * rng.standard_normal((H, W, S_total)))).astype(np.float32)
    cube[5:9, 3, 12] = hip_mag1c.NODATA
    cube[: H - 6, 5, 4] = hip_mag1c.NODATA                   # 6 valid pixels: skipped
    ids = np.array([1, 2, 3, 4, 5, 6, 7, 7, 8, 8, 9, 9, 10])  # widest run 2 columns: 2 x 256 = 512 pixels
    groups = np.repeat(ids[None, :], H, 0)
    x = torch.from_numpy(cube).to(DEV)
    mask = (rng.random((H, W)) > 0.05) & np.all(cube[..., 4:4 + S] > hip_mag1c.NODATA, axis=-1)
    sl = slice(4, 4 + S)
    for m in (None, mask):
        res = {}
        for on in (True, False):
            hip_mag1c.DIRECT_TILES = on
            try:
                res[on] = hip_mag1c.acrwl1mf_by_groups(x, t, groups, mask=m, alpha=alpha, band_slice=sl)
            finally:
                hip_mag1c.DIRECT_TILES = True
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
        assert bool((res[True][0][:, 5] == hip_mag1c.NODATA).all()) and bool((res[True][0][:, 0] != hip_mag1c.NODATA).any())
    # against the oracle (float64 on the same float32 radiances)
    want_mf, _ = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t, num_iter=30, alpha=alpha), cube[..., sl], groups)
    got = hip_mag1c.acrwl1mf_by_groups(x, t, groups, alpha=alpha, band_slice=sl)[0].cpu().numpy()
    assert np.array_equal(got == hip_mag1c.NODATA, want_mf == hip_mag1c.NODATA)
    ok = want_mf != hip_mag1c.NODATA
    assert np.mean(rel(got[ok], want_mf[ok]) > 1e-3) < 2e-3
    # an arbitrary (non-column) integer group map with every group below 512 pixels: the direct launch is tried and taken; with one
    # group above (ids2): it reports that group and the call is redone on the packed path -- both bit-identical to the packed path
    yy, xx = np.mgrid[0:H, 0:W]
    ids1 = ((yy // 64) * 7 + (xx * 5 + yy) % 7 + 1).astype(np.int64)            # 28 scattered groups of ~120 pixels
    ids2 = ids1.copy(); ids2[:48, :] = 99                                       # + one group of 624 pixels
    for ids_ in (ids1, ids2):
        res = {}
        for on in (True, False):
            hip_mag1c.DIRECT_TILES = on
            try:
                res[on] = hip_mag1c.acrwl1mf_by_groups(x, t, ids_, alpha=1e-3, band_slice=sl)
            finally:
                hip_mag1c.DIRECT_TILES = True
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
        assert bool((res[True][0] != hip_mag1c.NODATA).any())
    # a 3-column run (768 pixels) is beyond the register tile: the same call takes the packed path and still agrees with the oracle
    wide = np.repeat(np.array([1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5, 6])[None, :], H, 0)
    w_mf = hip_mag1c.acrwl1mf_by_groups(x, t, wide, alpha=alpha, band_slice=sl)[0].cpu().numpy()
    w_want, _ = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t, num_iter=30, alpha=alpha), cube[..., sl], wide)
    okw = w_want != hip_mag1c.NODATA
    assert np.array_equal(w_mf == hip_mag1c.NODATA, ~okw) and np.mean(rel(w_mf[okw], w_want[okw]) > 1e-3) < 2e-3


@pytest.mark.parametrize("step", [2, 4, None])
def test_emit_driver_vs_reference_golden(hip, step):
    """a14 pinned: the HIP EMIT driver (band selection, template from the shipped LUT, fill mask, column blocks, fp64 filter,
    float32 outputs) against what the reference's own mag1c_emit() returned for the same cube (golden G10)"""
    import g9_util
    g3, g10 = np.load(os.path.join(G, "g3_templates.npz")), np.load(os.path.join(G, "g10_emit_driver.npz"))
    seed, rows, cols = (int(v) for v in g10["meta"])
    raw = g9_util.emit_cube(seed, rows, cols, g3["emit_centers"])
    sel = (g3["emit_centers"] >= 2122) & (g3["emit_centers"] <= 2488)
    templ = hip_mag1c.generate_template_from_bands(g3["emit_centers"][sel], g3["emit_fwhm"][sel])[:, 1]
    mf, alb = hip_mag1c.mag1c_columns(torch.from_numpy(np.ascontiguousarray(raw[..., sel])).to(DEV), templ, -9999.0, column_step=step, num_iter=30)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    want_mf, want_alb = g10[f"mf_step{step}"], g10[f"albedo_step{step}"]
    assert mf.dtype == np.float32 and np.array_equal(mf == -9999.0, want_mf == -9999.0)
    ok = want_mf != -9999.0
    assert rel(mf[ok], want_mf[ok]).max() < 1e-5 and rel(alb[ok], want_alb[ok]).max() < 1e-6


def test_cfg3_size_properties(hip):
    """BASELINE config 3 shape (512 column groups x 512 px x 125 bands, fp32): size-independent properties --
    pixels without plume stay near zero, planted enhancement is recovered in order of magnitude, groups are
    independent (filtering a subset of the columns reproduces those columns bit for bit)."""
    rng = np.random.default_rng(8)
    S, H, W = 125, 512, 64                       # 64 of the 512 columns: the same kernel work per group
    t73 = np.load(os.path.join(G, "g3_templates.npz"))["aviris_template_kept"][:, 1]
    t = np.interp(np.linspace(0, 72, S), np.arange(73), t73)
    base = rng.uniform(1, 6, size=S)
    cube = base * (1 + 0.05 * rng.standard_normal((H, W, S)))
    conc = np.zeros((H, W)); conc[200:260, 10:30] = 2000.0
    cube = (cube * (1 + conc[..., None] * 1e-5 * t / 1.0)).astype(np.float32)     # exp(-k*c) ~ 1 + t*c*1e-5
    groups = np.arange(1, W + 1)[None, :].repeat(H, 0)
    x = torch.from_numpy(cube).to(DEV)
    mf, _ = hip_mag1c.acrwl1mf_by_groups(x, t, groups)
    mf = mf.cpu().numpy()
    assert np.isfinite(mf).all() and (mf >= 0).all()
    inside, outside = mf[200:260, 10:30], mf[:150]
    assert 1000 < np.median(inside) < 4000
    assert np.median(outside) < 50
    sub, _ = hip_mag1c.acrwl1mf_by_groups(x[:, 16:32].contiguous(), t, groups[:, 16:32])
    assert np.array_equal(sub.cpu().numpy(), mf[:, 16:32])


def test_cfg3_s125_vs_fp64_oracle(hip):
    """BASELINE configs[2] at its stated size per group: 125 bands, 512-pixel detector-column groups, fp32 radiances, alpha = 0,
    30 iterations (process_aviris.py:209-219 -> mag1c.py:177-280).  24 columns (24 x 31 Choleskys of 125^2 on the CPU) against
    oracle/mag1c_ref.acrwl1mf_group evaluated in float64 ON THE SAME float32 radiances (exact arithmetic = the truth): the HIP
    path computes in fp64, so every pixel must sit within 1e-5; and against the oracle run in the reference's own float32 within
    its fp32 noise (1e-3 on >= 99.8 % of the pixels).  NODATA pixels inside a group and a skipped group keep the exact pattern."""
    rng = np.random.default_rng(81)
    S, H, W = 125, 512, 24
    t73 = np.load(os.path.join(G, "g3_templates.npz"))["aviris_template_kept"][:, 1]
    t = np.interp(np.linspace(0, 72, S), np.arange(73), t73)
    base = rng.uniform(1, 6, size=S)
    cube = base * (1 + 0.05 * rng.standard_normal((H, W, S)))
    conc = np.zeros((H, W)); conc[200:260, 4:14] = 1500.0
    cube = (cube * (1 + conc[..., None] * 1e-5 * t)).astype(np.float32)
    cube[17:23, 5, 40] = hip_mag1c.NODATA                    # invalid pixels inside a group
    cube[: H - 9, 20, 0] = hip_mag1c.NODATA                  # 9 valid pixels: group skipped (<= 10)
    groups = np.arange(1, W + 1)[None, :].repeat(H, 0)
    mf, alb = hip_mag1c.acrwl1mf_by_groups(torch.from_numpy(cube).to(DEV), t, groups)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    want64, alb64 = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t, num_iter=30, alpha=0.0), cube.astype(np.float64), groups)
    want32, _ = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t.astype(np.float32), num_iter=30, alpha=0.0), cube, groups)
    assert np.array_equal(mf == hip_mag1c.NODATA, want64 == hip_mag1c.NODATA)
    assert (mf[:, 20] == hip_mag1c.NODATA).all() and (mf[17:23, 5] == hip_mag1c.NODATA).all()
    ok = want64 != hip_mag1c.NODATA
    e64, ealb = rel(mf[ok], want64[ok]), rel(alb[ok], alb64[ok])
    e32 = rel(mf[ok], want32[ok])
    print(f"configs[2] S=125 P=512 x {W} groups: mf vs fp64 oracle max {e64.max():.2e}, albedo {ealb.max():.2e}; "
          f"vs fp32 oracle: {np.mean(e32 > 1e-3) * 100:.3f} % of pixels over 1e-3 (max {e32.max():.2e})")
    assert e64.max() < 1e-5 and ealb.max() < 1e-6
    assert np.mean(e32 > 1e-3) < 2e-3


def test_emit_full_height_block_vs_fp64_oracle(hip):
    """one full-height EMIT block at the granule's real geometry (1280 rows, 49 bands kept of 2122-2488 nm, column_step 2,
    alpha 1e-4, fill -9999: mag1c_emit.py:50-90) against oracle/mag1c_ref.mag1c_columns (float64): <= 1e-6 on every pixel --
    the goldens (G10) hold 96-row cubes only, and at 2560 pixels per group the kernel walks 40 pixel tiles per band."""
    import g9_util
    g3 = np.load(os.path.join(G, "g3_templates.npz"))
    sel = (g3["emit_centers"] >= 2122) & (g3["emit_centers"] <= 2488)
    raw = g9_util.emit_cube(5150, 1280, 8, g3["emit_centers"])[..., sel]
    raw = np.ascontiguousarray(raw)
    raw[100:140, 2, :] = -9999.0                       # fill pixels inside a block
    raw[:, 5, 3] = -9999.0                             # a column that is all fill in one band -> invalid
    S = int(sel.sum())
    assert S >= 40
    templ = hip_mag1c.generate_template_from_bands(g3["emit_centers"][sel], g3["emit_fwhm"][sel])[:, 1]
    mf, alb = hip_mag1c.mag1c_columns(torch.from_numpy(raw).to(DEV), templ, -9999.0, column_step=2, num_iter=30)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    want_mf, want_alb = mag1c_ref.mag1c_columns(raw, np.asarray(templ, dtype=np.float64), -9999.0, column_step=2, num_iter=30, alpha=1e-4)
    assert np.array_equal(mf == -9999.0, want_mf == -9999.0)
    ok = want_mf != -9999.0
    e, ea = rel(mf[ok], want_mf[ok]).max(), rel(alb[ok], want_alb[ok]).max()
    print(f"EMIT block 1280 x 2 x {S}: mf {e:.2e}, albedo {ea:.2e} vs the fp64 oracle")
    assert e < 1e-6 and ea < 1e-6


def test_id_layout_equals_sorted_layout(hip):
    """orthorectified group maps (|GLT sample index|, process_aviris.py:211-217) are laid out by a stable counting sort on the
    device (sc_mag1c_layout_ids); bit-identical to the general torch sort path: scattered ids with gaps, groups of <= 10 pixels
    (skipped), masked pixels, NODATA pixels; ids the counting sort does not take (negative) fall back to the sort"""
    rng = np.random.default_rng(12)
    t = np.load(os.path.join(G, "g3_templates.npz"))["aviris_template_kept"][:, 1][:24]
    H, W, S = 90, 70, 24
    cube = (rng.uniform(1, 6, size=S) * (1 + 0.05 * rng.standard_normal((H, W, S)))).astype(np.float32)
    cube[5:9, 3, 2] = hip_mag1c.NODATA
    ids = ((np.arange(W)[None, :] * 3 + np.arange(H)[:, None] // 4) % 37) * 5 + 2          # ids 2, 7, ..., 182 (gaps), scattered
    ids[40:42, 10:13] = 400                                                                # a group of 6 pixels: skipped
    ids[:3, :] = 0
    mask = (rng.random((H, W)) > 0.1) & (ids != 0)
    x = torch.from_numpy(cube).to(DEV)
    for m, mg in ((mask, None), (mask, 400), (None, None)):
        a_mf, a_alb = hip_mag1c.acrwl1mf_by_groups(x, t, ids, mask=m, max_group=mg)
        try:
            hip_mag1c.COLUMN_FAST_PATH = False
            b_mf, b_alb = hip_mag1c.acrwl1mf_by_groups(x, t, ids, mask=m)
        finally:
            hip_mag1c.COLUMN_FAST_PATH = True
        assert torch.equal(a_mf, b_mf) and torch.equal(a_alb, b_alb)
        assert bool((a_mf[40:42, 10:13] == hip_mag1c.NODATA).all())
        if m is not None:
            assert bool((a_mf[:3] == hip_mag1c.NODATA).all()) and bool((a_mf[~torch.from_numpy(m).to(DEV)] == hip_mag1c.NODATA).all())
    neg = ids.copy(); neg[ids == 7] = -7
    c_mf, _ = hip_mag1c.acrwl1mf_by_groups(x, t, neg, mask=mask)
    assert torch.equal(c_mf, a_mf if False else hip_mag1c.acrwl1mf_by_groups(x, t, ids, mask=mask)[0])


@pytest.mark.parametrize("S", [20, 77, 125])
def test_paired_launch_mixed_group_sizes(hip, S):
    """alpha = 0, fp32: sc_mag1c_groups launches the register-resident kernel (groups of <= 512 pixels) and the streaming kernel
    (larger groups) side by side, and each group is taken by exactly one of them, decided on the device.  One call with group sizes
    on both sides of the boundary (600, 513 | 512, 511, S + 100, S + 30) and a skipped one (5 <= 10 pixels), an arbitrary (non-column)
    group map, NODATA pixels inside groups; band counts that fill 2, 5 and 8 of the 16 x 16 blocks of the factorisation.  Every
    pixel within 1e-5 of oracle/mag1c_ref in float64 on the same float32 radiances."""
    rng = np.random.default_rng(100 + S)
    sizes = [600, 513, 512, 511, S + 100, S + 30, 5]             # (a group needs more valid pixels than bands: else C is singular)
    n = sum(sizes)
    H, W = 1, n
    t = -np.abs(rng.standard_normal(S)) * 0.3
    base = rng.uniform(1, 6, size=S)
    cube = base * (1 + 0.05 * rng.standard_normal((H, W, S))) + 0.2 * rng.standard_normal((H, W, 1)) * base
    conc = np.zeros((H, W)); conc[0, ::7] = 1500.0
    cube = (cube * (1 + conc[..., None] * 1e-5 * t)).astype(np.float32)
    ids = np.repeat(np.arange(1, len(sizes) + 1), sizes)
    perm = rng.permutation(n)                                   # groups interleaved over the pixels: the counting-sort layout
    groups = ids[perm][None, :]
    cube[0, np.flatnonzero(groups[0] == 3)[:4], 1] = hip_mag1c.NODATA      # 512 -> 508 valid pixels
    cube[0, np.flatnonzero(groups[0] == 2)[:2], 0] = hip_mag1c.NODATA      # 513 -> 511: crosses to the resident kernel
    mf, alb = hip_mag1c.acrwl1mf_by_groups(torch.from_numpy(cube).to(DEV), t, groups)
    mf, alb = mf.cpu().numpy(), alb.cpu().numpy()
    want, walb = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, t, num_iter=30, alpha=0.0), cube.astype(np.float64), groups)
    assert np.array_equal(mf == hip_mag1c.NODATA, want == hip_mag1c.NODATA)
    assert (mf[groups == 7] == hip_mag1c.NODATA).all()
    ok = want != hip_mag1c.NODATA
    worst = {}
    for gid, sz in enumerate(sizes[:-1], 1):
        sel = ok & (groups == gid)
        worst[sz] = float(rel(mf[sel], want[sel]).max())
        assert worst[sz] < 1e-5 and rel(alb[sel], walb[sel]).max() < 1e-6, (sz, worst)
    print(f"S={S}: worst relative error per group size {worst}")


@pytest.mark.parametrize("S", [24, 73])
def test_compute_energy(hip, S):
    """compute_energy=True of rmf / acrwl1mf (mag1c.py:270-275, 337-343) on float32 radiances against oracle/mag1c_ref evaluated in
    float64 on the same radiances (pinned on the reference's own float64 numbers by golden G11; the reference's float32 value of the rmf
    term is inf -- its product of Cholesky diagonals underflows -- and its float32 iteration terms carry 1e-5..1e-2 of rounding noise,
    so the float64 evaluation is the meaningful target).  With and without a statistics mask, alpha = 0 (from the Woodbury scalars)
    and 1e-4 (one more substitution per iteration); 24 bands (<= 64: four band slots per lane) and 73 (eight).  The residual term is
    evaluated as s^T C^{-1} s, not as the sum of a P x P matrix: without a mask its rmf-stage value is exactly 0 where the reference
    returns cancellation noise, so that stage is compared through the log-determinant term that dominates it."""
    rng = np.random.default_rng(500 + S)
    P, B = 300, 2
    t = -np.abs(rng.standard_normal(S)) * 0.3
    base = rng.uniform(1, 6, size=S)
    x = (base * (1 + 0.05 * rng.standard_normal((B, P, S))) + 0.2 * rng.standard_normal((B, P, 1)) * base).astype(np.float32)
    mask = rng.uniform(size=P) > 0.3
    xd = torch.from_numpy(x).to(DEV)
    for alpha in (0.0, 1e-4):
        for mk in (None, mask):
            kw = {} if mk is None else {"mask": torch.from_numpy(mk).to(DEV)}
            okw = {} if mk is None else {"mask": mk}
            mf, R, e = hip_mag1c.rmf(xd, t, alpha=alpha, compute_energy=True, **kw)
            wmf, wR, we = mag1c_ref.rmf(x.astype(np.float64), t, alpha=alpha, compute_energy=True, **okw)
            assert rel(mf.cpu().numpy(), wmf).max() < 1e-5                       # unscaled, as the reference returns it there
            assert abs(float(e) - we) <= 2e-6 * abs(we), (S, alpha, mk is None, float(e), we)
            mf, R, el = hip_mag1c.acrwl1mf(xd, t, num_iter=6, alpha=alpha, compute_energy=True, **kw)
            wmf, wR, wel = mag1c_ref.acrwl1mf(x.astype(np.float64), t, num_iter=6, alpha=alpha, compute_energy=True, **okw)
            assert rel(mf.cpu().numpy(), wmf).max() < 1e-5
            got = np.array([float(v) for v in el]); want = np.array(wel)
            assert len(el) == 7 and np.max(np.abs(got - want) / np.abs(want)) < 2e-5, (S, alpha, mk is None, got, want)


def test_compute_energy_float64_vs_reference_golden(hip):
    """float64 radiances (k_mag1c_fast for alpha = 0, k_mag1c for alpha != 0) against golden G11 = the numbers the reference itself returned
    for compute_energy=True: the rmf scalar (dominated by the log-determinant term) and the five iteration terms, with and without a mask."""
    g = np.load(os.path.join(G, "g11_energy.npz"))
    x, t, m = g["x_f64"], g["t"], g["mask_f64"]
    xd = torch.from_numpy(x).to(DEV)
    worst = 0.0
    for at, alpha in (("a0", 0.0), ("a1e4", 1e-4)):
        for mt, mk in (("nomask", None), ("mask", m)):
            kw = {} if mk is None else {"mask": torch.from_numpy(mk).to(DEV)}
            mf, R, e = hip_mag1c.rmf(xd, t, alpha=alpha, compute_energy=True, **kw)
            assert rel(mf.cpu().numpy(), g[f"rmf_f64_{at}_{mt}_mf"]).max() < 1e-9
            want = float(g[f"rmf_f64_{at}_{mt}_e"])
            assert abs(float(e) - want) <= 1e-9 * abs(want), (at, mt, float(e), want)
            mf, R, el = hip_mag1c.acrwl1mf(xd, t, num_iter=5, alpha=alpha, compute_energy=True, **kw)
            assert rel(mf.cpu().numpy(), g[f"acr_f64_{at}_{mt}_mf"]).max() < 1e-8
            got, wantl = np.array([float(v) for v in el[1:]]), g[f"acr_f64_{at}_{mt}_e"]
            assert abs(float(el[0]) - float(g[f"acr_f64_{at}_{mt}_e0"])) <= 1e-9 * abs(want)
            err = float(np.max(np.abs(got - wantl) / np.abs(wantl)))
            worst = max(worst, err)
            assert err < 1e-8, (at, mt, got, wantl)
    print(f"compute_energy, float64 radiances vs the reference's own values: worst relative error of an iteration term {worst:.1e}")
