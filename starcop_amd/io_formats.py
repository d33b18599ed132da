"""On-disk formats of the STARCOP hot path, read without rasterio / GDAL / spectral (none of them is in the build image).

What the reference reads and writes on this path:
  * one single-band float32 (labels: uint8) cloud-optimised GeoTIFF per product per 512 x 512 sample folder --
    ``rasterio.open(f"{folder}/{product}.tif").read(window=window)`` in starcop/data/dataset.py:59-102, written by
    ``save_cog(v, path, profile={"BLOCKSIZE": 128})`` in starcop/data/sampling_dataset.py:332-355 -- and the mag1c / albedo
    outputs of starcop/process_aviris.py:209-232;
  * the AVIRIS-NG radiance cube and its GLT as ENVI files opened as BIP memmaps (``spectral.io.envi.open(hdr)
    .open_memmap(interleave='bip')``, process_aviris.py:183-187).

``read_tiff`` decodes classic (32-bit offset) TIFFs, tiled or stripped, 8/16/32/64-bit unsigned / signed / float samples,
chunky or planar multi-band, compression none / deflate / LZW with predictor 1, 2 or 3, either byte order, and reads only the
blocks a window touches (the first IFD = full resolution of a COG).  ``write_tiff`` writes tiled (BLOCKSIZE 128), deflate or
uncompressed files that carry the GeoTIFF tags handed to it, so a round trip keeps the georeferencing.  ``open_envi`` parses an
ENVI header and returns the cube as a (lines, samples, bands) memmap view plus wavelengths / fwhm.  ``load_tileset`` reads the
sample folders of a split into pinned host buffers and uploads them asynchronously into a ``ResidentTileSet``.
LZW and the predictors are undone by host C++ in libstarcop_hip.so (sc_tiff_lzw_decode / sc_tiff_unpredict): a Python loop
over LZW codes would take seconds per tile.
"""
import os
import re
import struct
import zlib
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4),
          10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8)}
GEO_TAGS = (33550, 33922, 34264, 34735, 34736, 34737, 42112, 42113)       # pixel scale, tiepoints, transform, geo keys, GDAL metadata / nodata


class TiffInfo:
    def __init__(self):
        self.width = self.height = 0
        self.bands = 1
        self.dtype = np.dtype("uint8")
        self.block = (0, 0)            # (rows, cols) of a tile, or (rows_per_strip, width)
        self.tiled = False
        self.offsets = self.counts = ()
        self.compression, self.predictor, self.planar = 1, 1, 1
        self.byteorder = "<"
        self.tags: Dict[int, tuple] = {}           # raw values of every tag: {tag: (type, values)}

    def geo_tags(self):
        return {t: v for t, v in self.tags.items() if t in GEO_TAGS}


def tiff_info(path) -> TiffInfo:
    with open(path, "rb") as f:
        head = f.read(8)
        bo = {b"II": "<", b"MM": ">"}.get(head[:2])
        if bo is None:
            raise ValueError(f"{path}: not a TIFF file")
        magic, ifd = struct.unpack(bo + "HI", head[2:8])
        if magic == 43:
            raise NotImplementedError(f"{path}: BigTIFF is not supported (STARCOP samples are classic TIFFs of ~1 MB)")
        if magic != 42:
            raise ValueError(f"{path}: bad TIFF magic {magic}")
        f.seek(ifd)
        n, = struct.unpack(bo + "H", f.read(2))
        raw = f.read(12 * n)
        info = TiffInfo()
        info.byteorder = bo
        for i in range(n):
            tag, typ, cnt, val = struct.unpack(bo + "HHI4s", raw[12 * i:12 * i + 12])
            if typ not in _TYPES:
                continue
            code, size = _TYPES[typ]
            nbytes = size * cnt
            if nbytes <= 4:
                data = val[:nbytes]
            else:
                off, = struct.unpack(bo + "I", val)
                pos = f.tell(); f.seek(off); data = f.read(nbytes); f.seek(pos)
            if typ == 2:
                vals = (data.rstrip(b"\0").decode("latin-1"),)
            else:
                vals = struct.unpack(bo + code[0] * (cnt * len(code)), data)
            info.tags[tag] = (typ, vals)
    t = info.tags

    def one(tag, default=None):
        return t[tag][1][0] if tag in t else default
    info.width, info.height = one(256), one(257)
    info.bands = one(277, 1)
    bits = t.get(258, (3, (1,)))[1]
    fmt = t.get(339, (3, (1,)))[1][0]
    if len(set(bits)) != 1:
        raise NotImplementedError(f"{path}: bands of different bit depth")
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits[0] not in (8, 16, 32, 64):
        raise NotImplementedError(f"{path}: sample format {fmt} with {bits[0]} bits")
    info.dtype = np.dtype(f"{bo}{kind}{bits[0] // 8}")
    info.compression, info.predictor, info.planar = one(259, 1), one(317, 1), one(284, 1)
    if 322 in t:
        info.tiled = True
        info.block = (one(323), one(322))
        info.offsets, info.counts = t[324][1], t[325][1]
    else:
        info.block = (min(one(278, info.height), info.height), info.width)
        info.offsets, info.counts = t[273][1], t[279][1]
    return info


def _decode_block(buf: bytes, info: TiffInfo, rows: int, cols: int, spp: int) -> np.ndarray:
    """one tile / strip -> (rows, cols, spp) array in native byte order"""
    n = rows * cols * spp * info.dtype.itemsize
    c = info.compression
    if c == 1:
        raw = buf
    elif c in (8, 32946):
        raw = zlib.decompress(buf)
    elif c == 5:
        from . import _lib
        raw = _lib.tiff_lzw_decode(buf, n)
    else:
        raise NotImplementedError(f"TIFF compression {c} (supported: none, deflate, LZW)")
    if len(raw) < n:
        raise ValueError(f"TIFF block decodes to {len(raw)} bytes, expected {n}")
    a = np.frombuffer(raw, dtype=np.uint8, count=n)
    if info.predictor == 3:          # floating-point predictor: bytes of a row are de-interleaved by significance and differenced
        from . import _lib
        a = _lib.tiff_unpredict(a, 3, rows, cols, spp, info.dtype.itemsize, info.byteorder == ">")
        return a.view(info.dtype.newbyteorder("=")).reshape(rows, cols, spp)
    a = a.view(info.dtype).reshape(rows, cols, spp).astype(info.dtype.newbyteorder("="))
    if info.predictor == 2:          # horizontal differencing, per sample, modular integer arithmetic
        if a.dtype.kind == "f":
            raise NotImplementedError("TIFF predictor 2 on floating-point samples")
        np.cumsum(a, axis=1, dtype=a.dtype, out=a)
    elif info.predictor != 1:
        raise NotImplementedError(f"TIFF predictor {info.predictor}")
    return a


def read_tiff(path, window: Optional[Tuple[int, int, int, int]] = None, info: Optional[TiffInfo] = None) -> np.ndarray:
    """-> (bands, h, w) array; ``window`` = (row_off, col_off, height, width) inside the image (what
    ``rasterio.windows.Window(col_off, row_off, width, height)`` selects in dataset.py:59-76); only the blocks the window touches
    are read and decoded."""
    info = info or tiff_info(path)
    r0, c0, h, w = (0, 0, info.height, info.width) if window is None else (int(v) for v in window)
    if r0 < 0 or c0 < 0 or h <= 0 or w <= 0 or r0 + h > info.height or c0 + w > info.width:
        raise ValueError(f"{path}: window {window} outside the {info.height} x {info.width} image")
    bh, bw = info.block
    nby, nbx = -(-info.height // bh), -(-info.width // bw)
    planes = info.bands if info.planar == 2 else 1
    spp = 1 if info.planar == 2 else info.bands
    out = np.empty((info.bands, h, w), dtype=info.dtype.newbyteorder("="))
    with open(path, "rb") as f:
        for pl in range(planes):
            for by in range(r0 // bh, (r0 + h - 1) // bh + 1):
                for bx in range(c0 // bw, (c0 + w - 1) // bw + 1):
                    k = pl * nby * nbx + by * nbx + bx
                    rows = bh if info.tiled else min(bh, info.height - by * bh)      # tiles are always full size, strips are not
                    if info.counts[k] == 0:
                        # GDAL sparse file (SPARSE_OK=TRUE): a block that was never written has offset = byte count = 0 and
                        # reads as the nodata value (GDAL_NODATA, tag 42113) or zeros
                        nod = info.tags.get(42113)
                        try:
                            fillv = float(nod[1][0].strip().strip("\0")) if nod else 0.0
                        except (ValueError, AttributeError, IndexError):
                            fillv = 0.0
                        blk = np.full((rows, bw, spp), fillv, dtype=out.dtype)
                    else:
                        f.seek(info.offsets[k])
                        blk = _decode_block(f.read(info.counts[k]), info, rows, bw, spp)
                    y0, x0 = by * bh, bx * bw
                    ys, ye = max(r0, y0), min(r0 + h, y0 + rows)
                    xs, xe = max(c0, x0), min(c0 + w, x0 + bw)
                    part = blk[ys - y0:ye - y0, xs - x0:xe - x0, :]
                    if info.planar == 2:
                        out[pl, ys - r0:ye - r0, xs - c0:xe - c0] = part[..., 0]
                    else:
                        out[:, ys - r0:ye - r0, xs - c0:xe - c0] = np.moveaxis(part, 2, 0)
    return out


def write_tiff(path, array, blocksize: int = 128, compress: Optional[str] = "deflate", extra_tags: Optional[Dict[int, tuple]] = None):
    """(bands, h, w) or (h, w) array -> tiled little-endian TIFF with BLOCKSIZE x BLOCKSIZE tiles (128: the reference's
    ``profile={"BLOCKSIZE": 128}``), deflate or no compression, chunky samples.  ``extra_tags`` = {tag: (type, values)} as in
    ``TiffInfo.tags`` -- pass ``info.geo_tags()`` of a source file to keep the GeoTIFF georeferencing."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("write_tiff: expected a (bands, h, w) or (h, w) array")
    kind = {"u": 1, "i": 2, "f": 3}.get(a.dtype.kind)
    if kind is None or a.dtype.itemsize not in (1, 2, 4, 8):
        raise ValueError(f"write_tiff: unsupported dtype {a.dtype}")
    if blocksize % 16:
        raise ValueError("write_tiff: TIFF tile sizes must be multiples of 16")
    comp = {None: 1, "none": 1, "deflate": 8}.get(compress)
    if comp is None:
        raise ValueError(f"write_tiff: compress must be None or 'deflate' (got {compress!r})")
    nb, H, W = a.shape
    a = np.ascontiguousarray(np.moveaxis(a, 0, 2)).astype(a.dtype.newbyteorder("<"))
    nby, nbx = -(-H // blocksize), -(-W // blocksize)
    blocks = []
    for by in range(nby):
        for bx in range(nbx):
            t = np.zeros((blocksize, blocksize, nb), dtype=a.dtype)
            part = a[by * blocksize:(by + 1) * blocksize, bx * blocksize:(bx + 1) * blocksize]
            t[:part.shape[0], :part.shape[1]] = part
            raw = t.tobytes()
            blocks.append(zlib.compress(raw, 6) if comp == 8 else raw)
    entries = {256: (4, (W,)), 257: (4, (H,)), 258: (3, (8 * a.dtype.itemsize,) * nb), 259: (3, (comp,)), 262: (3, (1,)),
               277: (3, (nb,)), 284: (3, (1,)), 322: (4, (blocksize,)), 323: (4, (blocksize,)), 339: (3, (kind,) * nb)}
    if nb > 1:
        entries[338] = (3, (0,) * (nb - 1))          # extra samples: unspecified
    for tag, tv in (extra_tags or {}).items():
        if tag not in entries and tag not in (324, 325, 273, 279, 278):
            entries[int(tag)] = tv
    ntags = len(entries) + 2
    pos = 8 + 2 + 12 * ntags + 4                     # header + IFD, then out-of-line values, then the tiles
    blobs = []

    def pack(typ, vals):
        if typ == 2:
            return vals[0].encode("latin-1") + b"\0"
        code, _ = _TYPES[typ]
        return struct.pack("<" + code[0] * len(vals), *vals)
    ool = {}
    for tag, (typ, vals) in entries.items():
        data = pack(typ, vals)
        if len(data) > 4:
            ool[tag] = (pos, data); pos += len(data) + (len(data) & 1)
    inline = len(blocks) == 1                        # a single LONG fits the IFD entry itself
    off_pos = pos; pos += 0 if inline else 4 * len(blocks)
    cnt_pos = pos; pos += 0 if inline else 4 * len(blocks)
    offsets = []
    for b in blocks:
        offsets.append(pos); pos += len(b) + (len(b) & 1)
    entries[324] = (4, tuple(offsets)); entries[325] = (4, tuple(len(b) for b in blocks))
    ool[324] = (off_pos, pack(*entries[324])); ool[325] = (cnt_pos, pack(*entries[325]))
    if inline:
        ool.pop(324); ool.pop(325)
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, 8))
        f.write(struct.pack("<H", ntags))
        for tag in sorted(entries):
            typ, vals = entries[tag]
            data = pack(typ, vals)
            cnt = len(data) if typ == 2 else len(vals) // len(_TYPES[typ][0])
            if tag in ool:
                f.write(struct.pack("<HHII", tag, typ, cnt, ool[tag][0]))
            else:
                f.write(struct.pack("<HHI4s", tag, typ, cnt, data.ljust(4, b"\0")))
        f.write(struct.pack("<I", 0))
        for tag in sorted(ool, key=lambda k: ool[k][0]):
            p, data = ool[tag]
            assert f.tell() == p, (tag, f.tell(), p)
            f.write(data + (b"\0" if len(data) & 1 else b""))
        for b in blocks:
            f.write(b + (b"\0" if len(b) & 1 else b""))


# ------------------------------------------------------------------------------------------------ ENVI
_ENVI_DTYPES = {1: "u1", 2: "i2", 3: "i4", 4: "f4", 5: "f8", 12: "u2", 13: "u4", 14: "i8", 15: "u8"}


def read_envi_header(path_hdr) -> Dict[str, object]:
    """ENVI .hdr -> dict (lower-case keys; brace lists become lists of strings)"""
    txt = open(path_hdr, "r", errors="replace").read()
    if not txt.lstrip().upper().startswith("ENVI"):
        raise ValueError(f"{path_hdr}: not an ENVI header")
    out = {}
    for m in re.finditer(r"^\s*([^=\n]+?)\s*=\s*(\{[^}]*\}|[^\n]*)", txt, re.M):
        k, v = m.group(1).strip().lower(), m.group(2).strip()
        if v.startswith("{"):
            v = [s.strip() for s in v[1:-1].replace("\n", " ").split(",") if s.strip()]
        out[k] = v
    return out


def open_envi(path, writable: bool = False):
    """``path`` = the data file or its ``.hdr``.  -> (cube, meta): ``cube`` is a numpy memmap VIEW of shape (lines, samples,
    bands) whatever the file's interleave (the reference's ``open_memmap(interleave='bip')``, process_aviris.py:183-187; for BIP
    files the view is contiguous), ``meta`` = {"wavelengths", "fwhm" (float64 arrays or None), "header"}."""
    hdr = path if path.endswith(".hdr") else (path + ".hdr" if os.path.exists(path + ".hdr") else os.path.splitext(path)[0] + ".hdr")
    h = read_envi_header(hdr)
    dat = path if not path.endswith(".hdr") else next((c for c in (path[:-4], path[:-4] + ".img", path[:-4] + ".dat", path[:-4] + ".lut")
                                                       if os.path.exists(c)), None)
    if dat is None:
        raise FileNotFoundError(f"no data file next to {hdr}")
    ns, nl, nb = int(h["samples"]), int(h["lines"]), int(h["bands"])
    dt = _ENVI_DTYPES.get(int(h["data type"]))
    if dt is None:
        raise NotImplementedError(f"{hdr}: ENVI data type {h['data type']}")
    dt = np.dtype((">" if int(h.get("byte order", 0)) else "<") + dt)
    il = str(h.get("interleave", "bsq")).lower()
    shape = {"bip": (nl, ns, nb), "bil": (nl, nb, ns), "bsq": (nb, nl, ns)}[il]
    mm = np.memmap(dat, dtype=dt, mode="r+" if writable else "r", offset=int(h.get("header offset", 0)), shape=shape)
    cube = {"bip": mm, "bil": mm.transpose(0, 2, 1), "bsq": mm.transpose(1, 2, 0)}[il]

    def floats(key):
        return np.array([float(v) for v in h[key]], dtype=np.float64) if key in h else None
    return cube, {"wavelengths": floats("wavelength"), "fwhm": floats("fwhm"), "header": h}


def envi_geo_tags(header: Dict[str, object]) -> Dict[int, tuple]:
    """GeoTIFF georeferencing tags from an ENVI header's ``map info`` (what rasterio's ``src.transform`` / ``src.crs`` carry from
    the radiance file into the mag1c product, process_aviris.py:179-181,222-226).  ``map info = {UTM, x_ref, y_ref, easting,
    northing, x_size, y_size, zone, North|South, datum, units=..., rotation=deg}`` becomes the affine transform GDAL's ENVI
    driver builds (rotation in degrees, counter-clockwise, applied to the pixel axes) stored as ModelTransformationTag (34264) --
    or ModelPixelScale (33550) + ModelTiepoint (33922) when there is no rotation -- and a GeoKeyDirectory (34735) with
    ProjectedCSTypeGeoKey = EPSG 326zz / 327zz for WGS-84 UTM or GeographicTypeGeoKey 4326 for ``Geographic Lat/Lon``.
    Returns {} when the header has no usable map info."""
    mi = header.get("map info")
    if not isinstance(mi, list) or len(mi) < 7:
        return {}
    proj = mi[0].strip().lower()
    try:
        xr, yr, e0, n0, px, py = (float(v) for v in mi[1:7])
    except ValueError:
        return {}
    rot = 0.0
    for v in mi[7:]:
        if v.lower().replace(" ", "").startswith("rotation="):
            rot = float(v.split("=")[1])
    import math
    r = -math.radians(rot)
    gt = (e0 - (xr - 1.0) * px, math.cos(r) * px, -math.sin(r) * px, n0 + (yr - 1.0) * py, -math.sin(r) * py, -math.cos(r) * py)
    tags: Dict[int, tuple] = {}
    if rot == 0.0:
        tags[33550] = (12, (px, py, 0.0))
        tags[33922] = (12, (0.0, 0.0, 0.0, gt[0], gt[3], 0.0))
    else:
        tags[34264] = (12, (gt[1], gt[2], 0.0, gt[0], gt[4], gt[5], 0.0, gt[3], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0))
    if proj == "utm" and len(mi) >= 9:
        zone = int(float(mi[7]))
        epsg = (32700 if mi[8].strip().lower().startswith("s") else 32600) + zone
        # header (version 1, revision 1.0, 3 keys), GTModelType = projected, GTRasterType = PixelIsArea, ProjectedCSType
        tags[34735] = (3, (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, epsg))
    elif proj.startswith("geographic"):
        tags[34735] = (3, (1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326))
    return tags


def gdal_metadata_tag(tags: Dict[str, object], descriptions: Sequence[str] = ()) -> Dict[int, tuple]:
    """GDAL_METADATA (42112): the XML in which GDAL / rasterio keep dataset tags and band descriptions -- what
    ``save_cog(..., descriptions=[...], tags={...})`` leaves in the reference's products (process_aviris.py:222-232)."""
    def esc(v):
        return str(v).replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")
    items = []
    for k, v in tags.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            v = "[" + " ".join(repr(float(x)) for x in np.asarray(v).ravel()) + "]"
        items.append(f'  <Item name="{esc(k)}">{esc(v)}</Item>')
    for i, d in enumerate(descriptions):
        items.append(f'  <Item name="DESCRIPTION" sample="{i}" role="description">{esc(d)}</Item>')
    return {42112: (2, ("<GDALMetadata>\n" + "\n".join(items) + "\n</GDALMetadata>",))}


# ------------------------------------------------------------------------------------------------ sample folders
def load_sample(folder: str, products: Sequence[str], window=None) -> np.ndarray:
    """(len(products), h, w) float32: ``torch.cat([rasterio.open(f"{folder}/{p}.tif").read(window=window) ...]).float()``
    of starcop/data/dataset.py:66-76"""
    return np.concatenate([read_tiff(os.path.join(folder, f"{p}.tif"), window).astype(np.float32) for p in products], axis=0)


def load_tileset(folders: Sequence[str], input_products: Sequence[str], output_products: Sequence[str] = ("labelbinary",),
                 weight_loss: Optional[str] = "weight_mag1c", ids: Optional[Sequence[str]] = None, device="cuda", workers: int = 8):
    """Reads the sample folders of a split (the ``folder`` column of the reference's train.csv / test.csv, datamodule.py:98-106)
    into PINNED host buffers with a pool of decoder threads (zlib releases the GIL) and uploads each tensor with one
    asynchronous copy -> ``datamodule.ResidentTileSet`` (tiles resident in HBM for the whole training run)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from .datamodule import ResidentTileSet
    folders = list(folders)
    first = load_sample(folders[0], input_products)
    H, W = first.shape[-2:]
    M = len(folders)
    pin = torch.cuda.is_available()
    groups = [("inputs", list(input_products)), ("outputs", list(output_products))] + ([("weight_loss", [weight_loss])] if weight_loss else [])
    host = {name: torch.empty((M, len(p), H, W), dtype=torch.float32, pin_memory=pin) for name, p in groups}

    def work(i):
        for name, prods in groups:
            host[name][i] = torch.from_numpy(load_sample(folders[i], prods))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(work, range(M)))
    dev = {name: t.to(device, non_blocking=True) for name, t in host.items()}
    if pin:
        torch.cuda.current_stream().synchronize()        # the pinned staging buffers may be released after this
    return ResidentTileSet(dev["inputs"], dev["outputs"], dev.get("weight_loss"), ids=ids or [os.path.basename(f.rstrip("/")) for f in folders],
                           device=device)
