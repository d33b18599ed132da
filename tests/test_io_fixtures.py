"""The format readers against files written by INDEPENDENT encoders (libhdf5 1.10.6 via h5py, tifffile / imagecodecs, libtiff via
Pillow; tests/golden/make_io_fixtures.py, run once with the image's Anaconda interpreter, which the product never imports).
The expected arrays regenerate here from the same numpy PCG64 seeds -- the fixtures hold encoded bytes only."""
import os

import numpy as np
import pytest

from starcop_amd import hdf5_reader as h5
from starcop_amd import io_formats as io

G = os.path.join(os.path.dirname(__file__), "golden", "io")
ROWS, COLS, BANDS = 40, 32, 285


def emit_arrays():
    rng = np.random.default_rng(20260929)
    wl = np.linspace(381.0, 2493.0, BANDS).astype(np.float32)
    fwhm = np.full(BANDS, 8.5, np.float32)
    base = (1.0 + 5.0 * rng.random(BANDS)).astype(np.float32)
    rad = (np.round(base * (1.0 + 0.1 * (rng.random((ROWS, COLS, BANDS), dtype=np.float32) - 0.5)) * 16) / 16).astype(np.float32)
    dip = np.exp(-0.5 * ((wl - 2300.0) / 60.0) ** 2).astype(np.float32)
    rad[10:26, 8:20, :] *= (1.0 - 0.03 * dip)
    rad[:5, :3, :] = -9999.0
    rad[38, 7, 250] = -9999.0
    glt_x = rng.integers(0, COLS + 1, (60, 70)).astype(np.int32)
    glt_y = rng.integers(0, ROWS + 1, (60, 70)).astype(np.int32)
    lon = (10.0 + 0.001 * np.arange(COLS)[None, :] + 0.0 * np.arange(ROWS)[:, None]).astype(np.float64)
    lat = (45.0 - 0.001 * np.arange(ROWS)[:, None] + 0.0 * np.arange(COLS)[None, :]).astype(np.float64)
    return wl, fwhm, rad, glt_x, glt_y, lon, lat


@pytest.mark.parametrize("name,sb", [("emit_l1b_like_sb0.nc", 0), ("emit_l1b_like_sb2.nc", 2), ("emit_l1b_like_sb3.nc", 3)])
def test_hdf5_reader_on_libhdf5_written_emit_like_granules(name, sb):
    """three HDF5 format generations of the same EMIT-L1B-like NetCDF-4 layout: symbol-table vs link-message groups (11 root
    members: past the compact limit), object headers v1 / v2, chunk B-tree v1 vs fixed-array index, shuffle + deflate (+ fletcher32)
    filters, fill values, never-written chunks, hyperslab reads"""
    wl, fwhm, rad, glt_x, glt_y, lon, lat = emit_arrays()
    with h5.H5File(os.path.join(G, name)) as f:
        assert f.superblock_version == sb
        assert set(f.keys("/")) == {"downtrack", "crosstrack", "bands", "ortho_y", "ortho_x", "radiance", "sensor_band_parameters", "location",
                                    "flat_field_update", "build_dcid", "orbit_number"}
        assert f.keys("sensor_band_parameters") == ["fwhm", "wavelengths"] and set(f.keys("location")) == {"lon", "lat", "glt_x", "glt_y", "elev"}
        r = f["radiance"]
        assert r.shape == (ROWS, COLS, BANDS) and r.dtype == np.dtype("<f4")
        assert float(r.attrs["_FillValue"]) == -9999.0 and float(r.fillvalue) == -9999.0 and r.attrs["units"] == "uW/cm^2/SR/nm"
        assert np.array_equal(r.read(), rad)
        assert np.array_equal(r.read((slice(3, 37), slice(5, 30), slice(240, 283))), rad[3:37, 5:30, 240:283])
        assert np.array_equal(r[7:9, :, 31:33], rad[7:9, :, 31:33])                      # a slab that straddles chunk borders
        assert np.array_equal(f["sensor_band_parameters/wavelengths"].read(), wl) and np.array_equal(f["sensor_band_parameters/fwhm"][...], fwhm)
        assert np.array_equal(f["location/glt_x"].read(), glt_x) and np.array_equal(f["location/glt_y"].read(), glt_y)
        assert np.array_equal(f["location/lon"].read(), lon) and np.array_equal(f["location/lat"].read(), lat)      # fletcher32 + deflate
        elev = f["location/elev"].read()                                                # allocated, never written: all fill
        assert elev.shape == (ROWS, COLS) and (elev == -9999.0).all()
        assert float(f["build_dcid"].read()) == 10.0 and f["build_dcid"].shape == ()
        assert "radiance" in f and "location/nothing" not in f
        with pytest.raises(KeyError):
            f["sensor_band_parameters"]
        assert f.attrs("/")["title"].startswith("EMIT L1B")


def test_hdf5_paged_fixed_array_index():
    """libver='latest' datasets with more than 2^10 chunks: the fixed-array chunk index is PAGED (data block = prefix, page bitmap,
    checksum, then pages with a checksum each); `sparse` leaves its middle page uninitialised (fill value there)  [ADVICE r3]"""
    rng = np.random.default_rng(4242)
    a = rng.integers(-30000, 30000, 2600 * 3).astype(np.int16)
    b = (rng.random(2500 * 2, dtype=np.float32) * 8).astype(np.float32)
    want = np.full(b.shape, -1.0, np.float32)
    want[:1400] = b[:1400]
    want[4400:] = b[4400:]
    with h5.H5File(os.path.join(G, "fixed_array_paged_sb3.h5")) as f:
        assert f.superblock_version == 3
        assert np.array_equal(f["plain"].read(), a) and np.array_equal(f["plain"][3070:3080], a[3070:3080])     # across the page seam
        assert np.array_equal(f["sparse"].read(), want)


def test_read_emit_l1b_band_window():
    """read_emit_l1b: what EMITImage + read_from_bands + load_raw hand mag1c_emit (mag1c_emit.py:40-48) -- the bands inside
    [2122, 2488] nm as a contiguous slice read chunk-wise, band centres / widths, the fill value, the GLT"""
    wl, fwhm, rad, glt_x, glt_y, _, _ = emit_arrays()
    d = h5.read_emit_l1b(os.path.join(G, "emit_l1b_like_sb0.nc"), wavelength_range=(2122, 2488))
    keep = (wl >= 2122) & (wl <= 2488)
    b0, b1 = d["band_slice"]
    assert (b0, b1) == (int(np.flatnonzero(keep)[0]), int(np.flatnonzero(keep)[-1]) + 1) and b1 - b0 == int(keep.sum()) >= 40
    assert d["radiance"].dtype == np.float32 and np.array_equal(d["radiance"], rad[..., b0:b1]) and d["radiance"].flags.c_contiguous
    assert np.allclose(d["wavelengths"], wl[keep]) and np.allclose(d["fwhm"], 8.5) and d["fill_value"] == -9999.0
    assert np.array_equal(d["glt_x"], glt_x)
    full = h5.read_emit_l1b(os.path.join(G, "emit_l1b_like_sb3.nc"), rows=slice(4, 20))
    assert np.array_equal(full["radiance"], rad[4:20])
    with pytest.raises(ValueError):
        h5.read_emit_l1b(os.path.join(G, "emit_l1b_like_sb0.nc"), wavelength_range=(5000, 6000))
    with pytest.raises(h5.H5Error):
        h5.H5File(os.path.join(G, "tiled_f32_deflate_pred3.tif"))


def tiff_arrays():
    rng = np.random.default_rng(77)
    f32 = np.round(rng.random((1, 200, 150), dtype=np.float32) * 100, 2).astype(np.float32)
    u16 = rng.integers(0, 4000, (3, 130, 170)).astype(np.uint16)
    u8 = rng.integers(0, 255, (4, 64, 48)).astype(np.uint8)
    return f32, u16, u8


def test_tiff_reader_on_tifffile_and_libtiff_written_files():
    """tiles and strips, deflate and LZW, horizontal and floating-point predictors, chunky and planar samples, big-endian: files from
    tifffile / imagecodecs and from libtiff (Pillow) -- not from this package's own writer"""
    f32, u16, u8 = tiff_arrays()
    cases = [("tiled_f32_deflate_pred3.tif", f32), ("tiled_u16_deflate_pred2_chunky.tif", u16), ("strips_u16_planar_deflate.tif", u16),
             ("strips_u8_bigendian_none.tif", u8), ("strips_f32_lzw_libtiff.tif", f32), ("strips_u16_lzw_pred2_libtiff.tif", u16[:1]),
             ("strips_rgb8_lzw_libtiff.tif", u8[:3])]
    for name, want in cases:
        p = os.path.join(G, name)
        info = io.tiff_info(p)
        got = io.read_tiff(p)
        if name == "strips_f32_lzw_libtiff.tif" and 339 not in info.tags:
            # this Pillow build writes 32-bit samples without a SampleFormat tag; the TIFF default is unsigned integer, so the
            # reader returns the same bits as uint32
            assert got.dtype == np.uint32
            got = got.view(np.float32)
        assert got.shape == want.shape and got.dtype == want.dtype, (name, got.shape, got.dtype)
        assert np.array_equal(got, want), name
        h, w = want.shape[1:]
        win = (h // 3, w // 4, h // 2, w // 2)
        assert np.array_equal(io.read_tiff(p, window=win).view(want.dtype), want[:, win[0]:win[0] + win[2], win[1]:win[1] + win[3]]), name
    assert io.tiff_info(os.path.join(G, "tiled_f32_deflate_pred3.tif")).block == (128, 128)
    assert io.tiff_info(os.path.join(G, "tiled_f32_deflate_pred3.tif")).predictor == 3


def test_hdf5_reader_rejects_damaged_files(tmp_path):
    """a truncated granule or a file that is not HDF5 raises an exception (never a crash or a silent empty read)"""
    src = os.path.join(G, "emit_l1b_like_sb0.nc")
    raw = open(src, "rb").read()
    cut = tmp_path / "cut.nc"
    cut.write_bytes(raw[:len(raw) // 3])
    with pytest.raises(Exception):
        with h5.H5File(str(cut)) as f:
            f["radiance"].read()
    junk = tmp_path / "junk.nc"
    junk.write_bytes(b"CDF\x01" + b"\0" * 4096)                   # a NetCDF-3 classic header: not HDF5
    with pytest.raises(h5.H5Error):
        h5.H5File(str(junk))
