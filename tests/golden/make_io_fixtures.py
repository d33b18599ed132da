"""Fixture files for the format readers, written by INDEPENDENT encoders: libhdf5 1.10.6 through h5py 3.3 and tifffile 2021.7 /
imagecodecs / Pillow 8.4 (libtiff) -- the Anaconda interpreter of this image, which the product never imports:

    /opt/conda/bin/python3.9 tests/golden/make_io_fixtures.py

Every array regenerates from a numpy PCG64 seed (tests/test_io_fixtures.py), so the files are the only thing stored.
emit_l1b_like_*.nc mimic the layout of an EMIT L1B radiance granule (EMIT_L1B_RAD_*.nc: NetCDF-4 = HDF5): root variable `radiance`
(downtrack, crosstrack, bands) float32, chunked + shuffle + deflate with _FillValue -9999, groups sensor_band_parameters
(wavelengths, fwhm) and location (glt_x, glt_y, lon, lat), dimension scales attached the way netCDF-C does; three HDF5 format
generations (superblock 0 / 2 / 3, symbol-table and link-message groups, B-tree v1 and fixed-array chunk indexes)."""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "io")
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


def write_emit(path, libver):
    import h5py
    wl, fwhm, rad, glt_x, glt_y, lon, lat = emit_arrays()
    with h5py.File(path, "w", libver=libver, track_order=True) as f:
        f.attrs["title"] = "EMIT L1B At-Sensor Calibrated Radiance Data 60 m V001 (synthetic fixture)"
        f.attrs["ncei_template_version"] = "NCEI_NetCDF_Swath_Template_v2.0"
        for k in range(12):                                   # netCDF global attributes: enough to leave the compact form
            f.attrs[f"global_attribute_{k}"] = "x" * (10 + k)
        dims = {}
        for name, n in (("downtrack", ROWS), ("crosstrack", COLS), ("bands", BANDS), ("ortho_y", 60), ("ortho_x", 70)):
            d = f.create_dataset(name, data=np.zeros(n, np.float32))
            d.make_scale(name)
            dims[name] = d
        for name in ("flat_field_update", "build_dcid", "orbit_number"):      # scalar variables: 11 root members, past the compact-link limit
            f.create_dataset(name, data=np.float32(len(name)))
        r = f.create_dataset("radiance", data=rad, chunks=(16, 32, 32), compression="gzip", compression_opts=6, shuffle=True, fillvalue=np.float32(-9999.0))
        r.attrs["_FillValue"] = np.float32(-9999.0)
        r.attrs["long_name"] = "Radiance Data"
        r.attrs["units"] = "uW/cm^2/SR/nm"
        for i, name in enumerate(("downtrack", "crosstrack", "bands")):
            r.dims[i].attach_scale(dims[name])
        g = f.create_group("sensor_band_parameters", track_order=True)
        g.create_dataset("wavelengths", data=wl).attrs["units"] = "nm"
        g.create_dataset("fwhm", data=fwhm).attrs["units"] = "nm"
        for ds in (g["wavelengths"], g["fwhm"]):
            ds.dims[0].attach_scale(dims["bands"])
        loc = f.create_group("location", track_order=True)
        loc.create_dataset("lon", data=lon, chunks=(16, 32), compression="gzip", fletcher32=True)
        loc.create_dataset("lat", data=lat, chunks=(16, 32), compression="gzip")
        loc.create_dataset("glt_x", data=glt_x, chunks=(30, 70), compression="gzip", shuffle=True, fillvalue=np.int32(0))
        loc.create_dataset("glt_y", data=glt_y, chunks=(30, 70), compression="gzip", shuffle=True, fillvalue=np.int32(0))
        loc.create_dataset("elev", shape=(ROWS, COLS), dtype="f4", chunks=(16, 32), fillvalue=np.float32(-9999.0))   # never written: all fill


def paged_arrays():
    rng = np.random.default_rng(4242)
    return rng.integers(-30000, 30000, 2600 * 3).astype(np.int16), (rng.random(2500 * 2, dtype=np.float32) * 8).astype(np.float32)


def write_paged(path):
    """libver='latest' datasets whose fixed-array chunk index is PAGED (more than 2^10 chunks): one plain, completely written; one
    filtered and written only in its first 700 and last 300 chunks, so that the middle page of the index is never initialised"""
    import h5py
    a, b = paged_arrays()
    with h5py.File(path, "w", libver="latest") as f:
        f.create_dataset("plain", data=a, chunks=(3,))                                     # 2600 chunks -> 3 pages
        d = f.create_dataset("sparse", shape=b.shape, dtype="f4", chunks=(2,), compression="gzip", compression_opts=1,
                             fillvalue=np.float32(-1.0))                                   # 2500 chunks -> 3 pages
        d[:1400] = b[:1400]
        d[4400:] = b[4400:]


def tiff_arrays():
    rng = np.random.default_rng(77)
    f32 = np.round(rng.random((1, 200, 150), dtype=np.float32) * 100, 2).astype(np.float32)
    u16 = rng.integers(0, 4000, (3, 130, 170)).astype(np.uint16)
    u8 = rng.integers(0, 255, (4, 64, 48)).astype(np.uint8)
    return f32, u16, u8


def write_tiffs():
    import tifffile
    f32, u16, u8 = tiff_arrays()
    # what rasterio writes for the STARCOP products: one band, 128 x 128 tiles, deflate / lzw, float predictor
    tifffile.imwrite(os.path.join(OUT, "tiled_f32_deflate_pred3.tif"), f32[0], tile=(128, 128), compression="zlib", predictor=True)
    tifffile.imwrite(os.path.join(OUT, "tiled_u16_deflate_pred2_chunky.tif"), np.moveaxis(u16, 0, 2), tile=(64, 64), compression="zlib", predictor=True,
                     photometric="rgb", planarconfig="contig")
    # LZW through libtiff (Pillow): this imagecodecs build has no LZW encoder
    from PIL import Image
    Image.fromarray(f32[0], mode="F").save(os.path.join(OUT, "strips_f32_lzw_libtiff.tif"), compression="tiff_lzw")
    Image.fromarray(u16[0], mode="I;16").save(os.path.join(OUT, "strips_u16_lzw_pred2_libtiff.tif"), compression="tiff_lzw", tiffinfo={317: 2})
    Image.fromarray(np.ascontiguousarray(np.moveaxis(u8[:3], 0, 2)), mode="RGB").save(os.path.join(OUT, "strips_rgb8_lzw_libtiff.tif"), compression="tiff_lzw")
    tifffile.imwrite(os.path.join(OUT, "strips_u16_planar_deflate.tif"), u16, rowsperstrip=16, compression="zlib", photometric="minisblack",
                     planarconfig="separate")
    tifffile.imwrite(os.path.join(OUT, "strips_u8_bigendian_none.tif"), np.moveaxis(u8, 0, 2), byteorder=">", rowsperstrip=10, photometric="rgb",
                     planarconfig="contig", extrasamples=["unspecified"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    write_emit(os.path.join(OUT, "emit_l1b_like_sb0.nc"), "earliest")
    write_emit(os.path.join(OUT, "emit_l1b_like_sb2.nc"), ("v108", "v108"))
    write_emit(os.path.join(OUT, "emit_l1b_like_sb3.nc"), "latest")
    write_paged(os.path.join(OUT, "fixed_array_paged_sb3.h5"))
    write_tiffs()
    for n in sorted(os.listdir(OUT)):
        print(n, os.path.getsize(os.path.join(OUT, n)))
