"""On-disk formats: the dependency-free TIFF / ENVI readers and the tile writer of starcop_amd/io_formats.py.
rasterio / GDAL are absent from the image, so the fixtures are produced here: by ``write_tiff`` itself (round trips) and by
independent encoders written in this file straight from the TIFF 6.0 specification (LZW, predictors 2 and 3, strips, planar
layout, big-endian) -- the layouts GDAL's COG driver emits for the reference's ``save_cog(..., profile={"BLOCKSIZE": 128})``."""
import os
import struct
import zlib

import numpy as np
import pytest

from starcop_amd import _lib, io_formats as io


def lzw_encode(data: bytes) -> bytes:
    """TIFF 6.0 LZW encoder (MSB-first codes, early change), as libtiff writes it"""
    out = bytearray()
    acc = nacc = 0

    def put(code, bits):
        nonlocal acc, nacc
        acc = (acc << bits) | code
        nacc += bits
        while nacc >= 8:
            out.append((acc >> (nacc - 8)) & 0xFF)
            nacc -= 8
        acc &= (1 << nacc) - 1
    table = {bytes([i]): i for i in range(256)}
    nxt, bits = 258, 9
    put(256, bits)
    w = b""
    for b in data:
        wb = w + bytes([b])
        if wb in table:
            w = wb
            continue
        put(table[w], bits)
        table[wb] = nxt
        nxt += 1
        if nxt > (1 << bits) - 1 and bits < 12:
            bits += 1
        if nxt >= 4094:
            put(256, bits)
            table = {bytes([i]): i for i in range(256)}
            nxt, bits = 258, 9
        w = bytes([b])
    if w:
        put(table[w], bits)
    put(257, bits)
    if nacc:
        out.append((acc << (8 - nacc)) & 0xFF)
    return bytes(out)


def raw_tiff(path, blocks, tags, bo="<"):
    """minimal TIFF writer for hand-made fixtures: `tags` = {tag: (type, values)}; block offsets/counts tags are filled in"""
    off_tag, cnt_tag = (324, 325) if 322 in tags else (273, 279)
    ntags = len(tags) + 2
    pos = 8 + 2 + 12 * ntags + 4
    codes = {3: "H", 4: "I", 12: "d"}
    ool, body = {}, b""
    entries = dict(tags)
    entries[off_tag] = (4, [0] * len(blocks)); entries[cnt_tag] = (4, [len(b) for b in blocks])
    sizes = {}
    for tag, (typ, vals) in entries.items():
        n = struct.calcsize(codes[typ]) * len(vals)
        if n > 4:
            sizes[tag] = pos; pos += n + (n & 1)
    offs = []
    for b in blocks:
        offs.append(pos); pos += len(b) + (len(b) & 1)
    entries[off_tag] = (4, offs)
    with open(path, "wb") as f:
        f.write((b"II" if bo == "<" else b"MM") + struct.pack(bo + "HI", 42, 8) + struct.pack(bo + "H", ntags))
        for tag in sorted(entries):
            typ, vals = entries[tag]
            data = struct.pack(bo + codes[typ] * len(vals), *vals)
            if tag in sizes:
                f.write(struct.pack(bo + "HHII", tag, typ, len(vals), sizes[tag])); ool[sizes[tag]] = data
            else:
                f.write(struct.pack(bo + "HHI", tag, typ, len(vals)) + data.ljust(4, b"\0"))
        f.write(struct.pack(bo + "I", 0))
        for p in sorted(ool):
            assert f.tell() == p
            f.write(ool[p] + (b"\0" if len(ool[p]) & 1 else b""))
        for b in blocks:
            f.write(b + (b"\0" if len(b) & 1 else b""))


def test_lzw_decoder_against_spec_encoder():
    rng = np.random.default_rng(0)
    for n, hi in ((1, 2), (10, 6), (1000, 6), (70000, 4), (70000, 256), (300000, 3)):
        d = rng.integers(0, hi, n).astype(np.uint8).tobytes()
        assert _lib.tiff_lzw_decode(lzw_encode(d), n) == d
    d = bytes(100000)                                            # one long run: exercises the code == next case at every width
    assert _lib.tiff_lzw_decode(lzw_encode(d), len(d)) == d
    with pytest.raises(ValueError):
        _lib.tiff_lzw_decode(b"\x80\x7f\xff\xff\xff", 100)       # Clear, then a code beyond the table


def test_write_read_round_trip_and_windows(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((1, 512, 512)).astype(np.float32)
    for comp in ("deflate", None):
        p = str(tmp_path / f"mag1c_{comp}.tif")
        io.write_tiff(p, a, blocksize=128, compress=comp)
        info = io.tiff_info(p)
        assert info.tiled and info.block == (128, 128) and (info.height, info.width, info.bands) == (512, 512, 1)
        assert info.dtype == np.dtype("<f4") and info.compression == (8 if comp else 1)
        assert np.array_equal(io.read_tiff(p), a)
        for win in ((0, 0, 128, 128), (100, 50, 130, 140), (384, 384, 128, 128), (511, 0, 1, 512), (7, 9, 300, 17)):
            r0, c0, h, w = win
            assert np.array_equal(io.read_tiff(p, win), a[:, r0:r0 + h, c0:c0 + w])
        with pytest.raises(ValueError):
            io.read_tiff(p, (400, 400, 128, 128))
    # odd sizes (edge tiles are padded on disk), uint8 labels, multi-band chunky
    lab = (rng.random((300, 200)) > 0.9).astype(np.uint8)
    p = str(tmp_path / "labelbinary.tif")
    io.write_tiff(p, lab)
    assert np.array_equal(io.read_tiff(p)[0], lab) and io.read_tiff(p).dtype == np.uint8
    rgba = rng.integers(0, 255, (4, 70, 90)).astype(np.uint8)
    io.write_tiff(p, rgba, blocksize=16)
    assert np.array_equal(io.read_tiff(p), rgba)
    assert np.array_equal(io.read_tiff(p, (3, 5, 40, 60)), rgba[:, 3:43, 5:65])


def test_geotiff_tags_round_trip(tmp_path):
    a = np.arange(64 * 64, dtype=np.float32).reshape(1, 64, 64)
    geo = {33550: (12, (3.0, 3.0, 0.0)), 33922: (12, (0.0, 0.0, 0.0, 587000.0, 4100000.0, 0.0)),
           34735: (3, (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32611)), 42113: (2, ("-9999",))}
    p, q = str(tmp_path / "a.tif"), str(tmp_path / "b.tif")
    io.write_tiff(p, a, extra_tags=geo)
    info = io.tiff_info(p)
    assert info.geo_tags() == geo
    io.write_tiff(q, io.read_tiff(p) * 2, extra_tags=info.geo_tags())        # mag1c output keeps the input's georeferencing
    assert io.tiff_info(q).geo_tags() == geo and np.array_equal(io.read_tiff(q), a * 2)


@pytest.mark.parametrize("layout", ["lzw_tiles", "lzw_pred3_tiles", "deflate_pred2_u16_strips", "planar_big_endian_strips", "deflate_pred3_tiles"])
def test_foreign_layouts(tmp_path, layout):
    """files as other writers (GDAL / libtiff) lay them out, encoded here from the TIFF 6.0 specification"""
    rng = np.random.default_rng(2)
    p = str(tmp_path / "x.tif")
    H, W, T = 200, 144, 64

    def tiles(arr, enc):
        out = []
        for by in range(-(-H // T)):
            for bx in range(-(-W // T)):
                t = np.zeros((T, T), arr.dtype)
                part = arr[by * T:(by + 1) * T, bx * T:(bx + 1) * T]
                t[:part.shape[0], :part.shape[1]] = part
                out.append(enc(t))
        return out

    def pred3(t):                      # floating-point predictor: byte planes MSB first, then byte differencing along the row
        b = t.astype(">f4").view(np.uint8).reshape(t.shape[0], t.shape[1], 4)
        planes = np.concatenate([b[:, :, k] for k in range(4)], axis=1)
        d = planes.copy()
        d[:, 1:] = planes[:, 1:] - planes[:, :-1]
        return d.tobytes()
    base = {256: (4, [W]), 257: (4, [H]), 262: (3, [1]), 277: (3, [1]), 284: (3, [1])}
    if layout in ("lzw_tiles", "lzw_pred3_tiles", "deflate_pred3_tiles"):
        a = (rng.standard_normal((H, W)) * np.linspace(0.1, 50, W)).astype(np.float32)
        comp = 5 if layout.startswith("lzw") else 8
        pred = 3 if "pred3" in layout else 1
        packer = lzw_encode if comp == 5 else zlib.compress
        blocks = tiles(a, (lambda t: packer(pred3(t))) if pred == 3 else (lambda t: packer(t.tobytes())))
        raw_tiff(p, blocks, {**base, **{258: (3, [32]), 259: (3, [comp]), 317: (3, [pred]), 322: (4, [T]), 323: (4, [T]), 339: (3, [3])}})
        want = a[None]
    elif layout == "deflate_pred2_u16_strips":
        a = rng.integers(0, 4000, (H, W)).astype(np.uint16)
        rps = 37
        blocks = []
        for y in range(0, H, rps):
            s = a[y:y + rps].copy()
            d = s.copy(); d[:, 1:] = s[:, 1:] - s[:, :-1]
            blocks.append(zlib.compress(d.tobytes()))
        raw_tiff(p, blocks, {**base, **{258: (3, [16]), 259: (3, [8]), 317: (3, [2]), 278: (4, [rps]), 339: (3, [1])}})
        want = a[None]
    else:
        a = rng.standard_normal((2, H, W)).astype(np.float32)
        rps = 50
        blocks = [a[b, y:y + rps].astype(">f4").tobytes() for b in range(2) for y in range(0, H, rps)]
        raw_tiff(p, blocks, {**base, **{258: (3, [32, 32]), 259: (3, [1]), 277: (3, [2]), 284: (3, [2]), 278: (4, [rps]), 339: (3, [3, 3])}}, bo=">")
        want = a
    got = io.read_tiff(p)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    assert np.array_equal(io.read_tiff(p, (60, 30, 100, 90)), want[:, 60:160, 30:120])


def test_envi_bip_bil_bsq(tmp_path):
    rng = np.random.default_rng(3)
    nl, ns, nb = 11, 7, 5
    cube = rng.standard_normal((nl, ns, nb)).astype(np.float32)
    wl = np.linspace(2100.0, 2500.0, nb)
    for il, arr in (("bip", cube), ("bil", cube.transpose(0, 2, 1)), ("bsq", cube.transpose(2, 0, 1))):
        base = str(tmp_path / f"ang_{il}_img")
        np.ascontiguousarray(arr).astype("<f4").tofile(base)
        with open(base + ".hdr", "w") as f:
            f.write(f"ENVI\ndescription = {{test}}\nsamples = {ns}\nlines = {nl}\nbands = {nb}\nheader offset = 0\ndata type = 4\n"
                    f"interleave = {il}\nbyte order = 0\nwavelength = {{ {', '.join(f'{v:.3f}' for v in wl)} }}\n"
                    f"fwhm = {{ {', '.join(['5.6'] * nb)} }}\n")
        got, meta = io.open_envi(base)
        assert got.shape == (nl, ns, nb) and np.array_equal(np.asarray(got), cube)
        assert np.allclose(meta["wavelengths"], wl, atol=1e-3) and np.allclose(meta["fwhm"], 5.6)
        got2, _ = io.open_envi(base + ".hdr")
        assert np.array_equal(np.asarray(got2[..., 1:4]), cube[..., 1:4])          # the band slice process_aviris.py:199-207 reads
    # GLT: 2-band int32 BIP, big-endian
    glt = rng.integers(-50, 50, (nl, ns, 2)).astype(">i4")
    base = str(tmp_path / "ang_glt")
    glt.tofile(base)
    with open(base + ".hdr", "w") as f:
        f.write(f"ENVI\nsamples = {ns}\nlines = {nl}\nbands = 2\nheader offset = 0\ndata type = 3\ninterleave = bip\nbyte order = 1\n")
    g, meta = io.open_envi(base)
    assert np.array_equal(np.asarray(g), glt) and meta["wavelengths"] is None


def test_ch4_lut_reads_through_open_envi():
    """the shipped CH4 look-up table is an ENVI BSQ float64 file: the generic reader and mag1c's own agree"""
    from starcop_amd import mag1c
    d = os.path.join(os.path.dirname(mag1c.__file__), "data")
    cube, meta = io.open_envi(os.path.join(d, "ch4.hdr"))
    rads, wave = mag1c.read_ch4_lut()
    assert np.array_equal(np.asarray(cube).squeeze(), rads) and np.array_equal(meta["wavelengths"], wave)


def test_load_sample_matches_dataset_semantics(tmp_path):
    """dataset.py:66-76: per product one single-band file, read with the same window, concatenated on the band axis, .float()"""
    rng = np.random.default_rng(4)
    folder = tmp_path / "ang2019_sample"
    folder.mkdir()
    prods = {"mag1c": rng.uniform(0, 3000, (512, 512)).astype(np.float32), "TOA_AVIRIS_640nm": rng.uniform(5, 110, (512, 512)).astype(np.float32),
             "TOA_AVIRIS_550nm": rng.uniform(5, 110, (512, 512)).astype(np.float32), "TOA_AVIRIS_460nm": rng.uniform(5, 110, (512, 512)).astype(np.float32),
             "labelbinary": (rng.random((512, 512)) > 0.95).astype(np.uint8)}
    for k, v in prods.items():
        io.write_tiff(str(folder / f"{k}.tif"), v)
    names = ["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"]
    x = io.load_sample(str(folder), names)
    assert x.shape == (4, 512, 512) and x.dtype == np.float32 and all(np.array_equal(x[i], prods[n]) for i, n in enumerate(names))
    win = (200, 64, 128, 128)
    y = io.load_sample(str(folder), ["labelbinary"], win)
    assert y.dtype == np.float32 and np.array_equal(y[0], prods["labelbinary"][200:328, 64:192].astype(np.float32))


def test_sparse_tiles_read_as_nodata(tmp_path):
    """GDAL SPARSE_OK files: a tile that was never written has offset = byte count = 0 and reads as GDAL_NODATA (or zeros)"""
    import struct
    a = np.arange(256 * 256, dtype=np.float32).reshape(256, 256)
    p = str(tmp_path / "sparse.tif")
    io.write_tiff(p, a, blocksize=128, compress="deflate", extra_tags={42113: (2, ("-9999.0",))})
    info = io.tiff_info(p)
    raw = bytearray(open(p, "rb").read())
    # zero the offset and byte count of tile 2 (second row, first column) inside the out-of-line TileOffsets / TileByteCounts arrays
    n = struct.unpack_from("<H", raw, 8)[0]
    for k in range(n):
        tag, typ, cnt, val = struct.unpack_from("<HHII", raw, 10 + 12 * k)
        if tag in (324, 325):
            struct.pack_into("<I", raw, val + 4 * 2, 0)
    open(p, "wb").write(bytes(raw))
    back = io.read_tiff(p)[0]
    assert (back[128:, :128] == -9999.0).all()
    assert np.array_equal(back[:128], a[:128]) and np.array_equal(back[128:, 128:], a[128:, 128:])
    assert (io.read_tiff(p, window=(100, 100, 60, 60))[0][28:, :28] == -9999.0).all()


def test_envi_map_info_to_geotiff_tags(tmp_path):
    """the georeferencing run_mag1c copies from the radiance file (process_aviris.py:179-181): ENVI map info -> GeoTIFF tags"""
    hdr = {"map info": ["UTM", "1", "1", "500000.0", "4100000.0", "5.0", "5.0", "11", "North", "WGS-84", "units=Meters"]}
    t = io.envi_geo_tags(hdr)
    assert t[33550] == (12, (5.0, 5.0, 0.0)) and t[33922] == (12, (0.0, 0.0, 0.0, 500000.0, 4100000.0, 0.0))
    assert t[34735][1][-1] == 32611 and t[34735][1][3] == 3
    hdr["map info"] = hdr["map info"] + ["rotation=30.0"]
    hdr["map info"][8] = "South"
    t = io.envi_geo_tags(hdr)
    m = t[34264][1]
    assert 33550 not in t and t[34735][1][-1] == 32711
    c, s_ = np.cos(np.radians(30.0)), np.sin(np.radians(30.0))
    assert np.allclose([m[0], m[1], m[3], m[4], m[5], m[7]], [5 * c, 5 * s_, 500000.0, 5 * s_, -5 * c, 4100000.0])
    assert io.envi_geo_tags({}) == {} and io.envi_geo_tags({"map info": ["UTM", "x"]}) == {}
    # tags survive a write / read cycle together with the GDAL metadata the reference's save_cog leaves
    p = str(tmp_path / "g.tif")
    md = io.gdal_metadata_tag({"wavelengths": np.array([2122.5, 2127.5]), "mag1c": "acfwl1mf"}, ["CH4 Absorption (ppm x m)"])
    io.write_tiff(p, np.zeros((32, 32), np.float32), blocksize=16, extra_tags={**t, **md})
    info = io.tiff_info(p)
    assert info.tags[34264][1] == m and info.tags[34735][1][-1] == 32711
    xml = info.tags[42112][1][0]
    assert '<Item name="mag1c">acfwl1mf</Item>' in xml and 'role="description">CH4 Absorption (ppm x m)</Item>' in xml and "2122.5" in xml
