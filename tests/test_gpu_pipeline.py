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
    # throughput mode: full-width row strips (vertical halo only) -- the same exactness at (512 + 640) / 512 = 2.25x the scene's work
    # instead of 5x for square cores
    rs = pipeline.scene_tiles(H, W, 512, 320, strips=True)
    assert rs.shape[0] == 2 and rs[:, 2].tolist() == [0, 0] and rs[:, 3].tolist() == [W, W] and rs[1, 4:6].tolist() == [192, 1024]
    strips = pipeline.tiled_logits(model, x, tile=512, halo=pipeline.RECEPTIVE_HALO, strips=True)
    es = relerr(strips, whole)
    print(f"strips (512 rows + halo 320) vs whole-scene logits: rel err {es:.2e}")
    assert es < 1e-4 and torch.equal((strips >= 0)[far], (whole >= 0)[far])


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


def test_aviris_scene_mag1c_from_envi_files(hip, tmp_path):
    """process_aviris.run_mag1c (:146-232) end to end from ENVI files on disk: BIP radiance + GLT memmaps -> band selection ->
    CH4 target from the shipped LUT -> acrwl1mf per detector sample (|GLT sample index|, 0 = no data) -> tiled GeoTIFF output;
    against the oracle's func_by_groups on the same arrays, and the written file read back"""
    import os
    from oracle import mag1c_ref
    from starcop_amd import io_formats as io, mag1c
    rng = np.random.default_rng(5)
    nl, ns = 96, 40
    wl = np.linspace(2000.0, 2500.0, 101)                     # 5 nm grid: 74 bands inside [2122, 2488] and outside the bad-band windows
    fwhm = np.full_like(wl, 5.6)
    target = mag1c.generate_template_from_bands(wl, fwhm)[:, 1]
    cube = (rng.uniform(1, 6, size=wl.size) * (1 + 0.05 * rng.standard_normal((nl, ns, wl.size)))).astype(np.float32)
    k = np.zeros((nl, ns)); k[30:50, 10:20] = 2e-5
    cube = (cube * (1 + k[..., None] * np.nan_to_num(target))).astype(np.float32)
    glt = np.zeros((nl, ns, 2), dtype=np.int32)
    glt[..., 0] = (np.arange(ns)[None, :] + (np.arange(nl)[:, None] // 8)) % 23 + 1      # orthorectified: detector sample varies along a row AND down
    glt[:5, :, 0] = 0                                                                    # no-data border
    glt[..., 0] *= np.where(rng.random((nl, ns)) < 0.5, 1, -1)                           # interpolated pixels carry a negative index
    glt[..., 1] = np.arange(nl)[:, None]
    folder = tmp_path / "ang20190101t000000_rdn"
    folder.mkdir()
    name = folder.name
    for suffix, arr, dt, extra in (("img", cube, 4, f"wavelength = {{ {', '.join(f'{v:.4f}' for v in wl)} }}\nfwhm = {{ {', '.join(f'{v:.2f}' for v in fwhm)} }}\n"
                                    "map info = {UTM, 1, 1, 500000.0, 4100000.0, 5.0, 5.0, 11, North, WGS-84, units=Meters, rotation=-12.0}\n"),
                                   ("glt", glt, 3, "")):
        arr.tofile(str(folder / f"{name}_{suffix}"))
        (folder / f"{name}_{suffix}.hdr").write_text(f"ENVI\nsamples = {ns}\nlines = {nl}\nbands = {arr.shape[2]}\nheader offset = 0\n"
                                                     f"data type = {dt}\ninterleave = bip\nbyte order = 0\n{extra}")
    out_mf, out_alb = str(tmp_path / "mag1c.tif"), str(tmp_path / "albedo.tif")
    mf, alb = pipeline.aviris_scene_mag1c(str(folder), out_mf, out_alb)
    keep = mag1c.get_mask_bad_bands(wl) & (wl >= 2122) & (wl <= 2488)
    groups = np.abs(glt[..., 0])
    want_mf, want_alb = mag1c_ref.func_by_groups(lambda xg: mag1c_ref.acrwl1mf_group(xg, target[keep], num_iter=30, alpha=0.0),
                                                 cube[..., keep], groups, mask=groups != 0)
    got = mf.cpu().numpy()
    assert np.array_equal(got == mag1c.NODATA, want_mf == mag1c.NODATA) and (got[:5] == mag1c.NODATA).all()
    ok = want_mf != mag1c.NODATA
    d = np.abs(got[ok] - want_mf[ok]) / max(float(np.abs(want_mf[ok]).max()), 1.0)
    assert np.mean(d > 1e-3) < 2e-3
    back = io.read_tiff(out_mf)
    info = io.tiff_info(out_mf)
    assert np.array_equal(back[0], got) and info.block == (128, 128) and info.tags[42113][1][0] == str(mag1c.NODATA)
    assert np.array_equal(io.read_tiff(out_alb)[0], alb.cpu().numpy())
    # what the reference's save_cog call leaves in the product: transform + crs of the radiance file, description, tags
    assert info.tags[34735][1][-1] == 32611 and abs(info.tags[34264][1][3] - 500000.0) < 1e-9
    xml = info.tags[42112][1][0]
    assert '<Item name="mag1c">acfwl1mf</Item>' in xml and "CH4 Absorption (ppm x m)" in xml and repr(float(wl[keep][0])) in xml
    assert "Albedo" in io.tiff_info(out_alb).tags[42112][1][0]


def test_emit_granule_predict_from_netcdf_file(hip):
    """the notebook path from the FILE.  An EMIT-L1B-like NetCDF-4 granule written by libhdf5 (tests/golden/io,
    make_io_fixtures.py) -> own HDF5 reader (only the mag1c band slice and the RGB planes are read) -> template from the file's
    band centres / widths -> mag1c -> rescale -> U-Net -> mask; equal to emit_scene_predict on the array a full read returns,
    and mf against the fp64 oracle"""
    import os
    from starcop_amd import hdf5_reader, mag1c
    path = os.path.join(os.path.dirname(__file__), "golden", "io", "emit_l1b_like_sb0.nc")
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(DEV).eval()
    out = pipeline.emit_granule_predict(model, path, column_step=4, ratio_bands=(2350, 2310))
    full = hdf5_reader.read_emit_l1b(path)                              # all 285 bands
    wl = full["wavelengths"]
    keep = (wl >= 2122) & (wl <= 2488)
    templ = mag1c.generate_template_from_bands(wl[keep], full["fwhm"][keep])
    want = pipeline.emit_scene_predict(model, full["radiance"], wl, templ, fill_value=full["fill_value"], column_step=4, ratio_bands=(2350, 2310))
    for k in ("mf", "albedo", "input", "prediction", "pred_binary", "ratio"):
        assert torch.equal(out[k], want[k]), k
    assert np.allclose(out["wavelengths"], wl[keep]) and out["glt_x"].shape == (60, 70) and out["fill_value"] == -9999.0
    assert out["prediction"].shape == (32, 32) and out["mf"].shape == (40, 32)
    sub = full["radiance"][..., keep]
    mf_ref, _ = mag1c_ref.mag1c_columns(sub, templ[:, 1], -9999.0, column_step=4)
    assert np.array_equal(out["mf"].cpu().numpy() == -9999.0, mf_ref == -9999.0) and bool((out["mf"][:5, :3] == -9999.0).all())
    ok = mf_ref != -9999.0
    assert np.abs(out["mf"].cpu().numpy()[ok] - mf_ref[ok]).max() < 1e-4 * max(1.0, float(np.abs(mf_ref[ok]).max()))


def test_cfg5_full_size_scene(hip):
    """BASELINE configs[4] at ITS size -- the 1280 x 1242 x 285 EMIT-like cube bench.py's `extra.emit_scene` times (notebook
    inference_on_raw_EMIT_nc_file.ipynb cells 8-19) -- checked stage by stage, not only timed:
      * mf / albedo of 8 column blocks spread over the scene (incl. the ragged last one and one with fill pixels) against
        oracle/mag1c_ref.mag1c_columns (fp64 numpy, 1280-pixel x 2-column groups, 31 covariance rounds each);
      * the two-band ratio on the full 1280 x 1242 planes against oracle/host_ref.band_ratio (exact percentiles of 1.6 M values);
      * the network input against host_ref.emit_rescale;
      * logits of the throughput mode (512-row full-width strips + 320-px halo, what each rank of a tile-sharded job runs)
        against the whole-scene forward: <= 1e-5 of the logit range; masks equal away from logit 0."""
    dev = DEV
    wl = np.linspace(381.0, 2493.0, 285)
    keep = np.nonzero((wl >= 2122.0) & (wl <= 2488.0))[0]
    rng = np.random.default_rng(5)
    templ = -np.abs(rng.standard_normal(keep.size)) * 0.3 - 0.05
    gen = torch.Generator(device=dev).manual_seed(11)
    rows, cols = 1280, 1242
    cube = ((torch.rand(285, generator=gen, device=dev) * 5 + 1) * (1 + 0.05 * torch.randn(rows, cols, 285, generator=gen, device=dev))).float()
    cube[100:140, 600:602, :] = -9999.0                       # fill pixels inside one of the checked column blocks
    cube[:30, :7, :] = -9999.0
    cube = cube.contiguous()
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()
    whole = pipeline.emit_scene_predict(model, cube, wl, templ, column_step=2, ratio_bands=(2350, 2310))
    strips = pipeline.emit_scene_predict(model, cube, wl, templ, column_step=2, ratio_bands=(2350, 2310), tile=512)
    assert whole["prediction"].shape == (1280, 1216) and whole["mf"].shape == (rows, cols)
    # -- matched filter: 8 column blocks against the fp64 oracle
    sub = cube[..., int(keep[0]):int(keep[-1]) + 1]
    worst = 0.0
    for c0 in (0, 2, 310, 600, 620, 900, 1238, 1240):
        blk = sub[:, c0:c0 + 2].cpu().numpy()
        mf_ref, alb_ref = mag1c_ref.mag1c_columns(blk, templ, -9999.0, column_step=2)
        for got, want in ((whole["mf"][:, c0:c0 + 2], mf_ref), (whole["albedo"][:, c0:c0 + 2], alb_ref)):
            want = torch.as_tensor(want)
            assert torch.equal(got.cpu() == -9999.0, want == -9999.0), c0         # exact fill pattern
            worst = max(worst, relerr(got, want))
    print(f"configs[4] full size: mf / albedo of 8 column blocks vs the fp64 oracle: worst rel err {worst:.2e}")
    assert worst < 1e-4
    assert torch.equal(whole["mf"], strips["mf"])
    # -- band ratio on the scene-sized planes
    ia, ir = pipeline.nearest_bands(wl, (2350, 2310))
    r_ref = host_ref.band_ratio(cube[..., ia].cpu().numpy(), cube[..., ir].cpu().numpy())
    e_r = relerr(whole["ratio"], torch.as_tensor(r_ref))
    print(f"configs[4] full size: band ratio vs host_ref.band_ratio: rel err {e_r:.2e}")
    assert e_r < 1e-4
    # -- network input
    rgb = cube[..., pipeline.nearest_bands(wl)].permute(2, 0, 1).cpu().numpy()
    x_ref = host_ref.emit_rescale(whole["mf"].cpu().numpy(), rgb)
    assert relerr(whole["input"], torch.as_tensor(x_ref)) < 1e-6
    # -- strips vs whole-scene logits (from the probabilities' logits: recompute both with the model for the raw logits)
    with torch.no_grad():
        lw = model(whole["input"][None])[0, 0]
    ls = pipeline.tiled_logits(model, whole["input"], tile=512, halo=pipeline.RECEPTIVE_HALO, strips=True)
    e = relerr(ls, lw)
    print(f"configs[4] full size: 512-row strips + halo 320 vs whole-scene logits: rel err {e:.2e}")
    assert e < 1e-5
    far = lw.abs() > 1e-3
    assert torch.equal(strips["pred_binary"][far], whole["pred_binary"][far])
