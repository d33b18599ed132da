"""End-to-end scene inference on the GPU: radiance cube -> mag1c -> network input -> plume mask.

The reference spreads this over its drivers and notebooks; the steps and their order are
  * EMIT (BASELINE configs[4]): ``mag1c_emit`` (starcop/models/mag1c_emit.py:16-90: bands in [2122, 2488] nm, float64
    filter on blocks of ``column_step`` columns) -> RGB = nearest bands to 640/550/460 nm -> range rescale of
    starcop/emit_tools/emit_dataset.py:62-106 -> ``model(x)`` -> sigmoid > 0.5;
  * AVIRIS-NG (configs[2]): ``run_mag1c`` (starcop/process_aviris.py:189-219: per detector column, alpha = 0) and the
    pre-computed RGB products -> ``padded_predict``.
Everything stays on the device between the stages; with ``torch.distributed`` initialised the column blocks of the
matched filter and the scenes are independent work items (``column_range`` / ``parallel.sharded_map``): no collective on
the data path.
"""
import numpy as np
import torch

from . import mag1c
from .features import emit_to_aviris_input
from .model_module import masks_from_logits

EMIT_MAG1C_RANGE_NM = (2122.0, 2488.0)          # mag1c_emit.py:40-43
RGB_NM = (640.0, 550.0, 460.0)


def nearest_bands(wavelengths, targets=RGB_NM):
    w = np.asarray(wavelengths, dtype=np.float64)
    return [int(np.argmin(np.abs(w - t))) for t in targets]


# Receptive field of smp.Unet('mobilenet_v2'): encoder 3x3 convolutions at strides 1..32 reach 490 px, the decoder
# adds 126 -> a logit depends on inputs at most 308 px away.  Halo >= 320 (a multiple of 32, the network's stride) therefore makes a
# tile's core logits IDENTICAL to the whole-scene forward in eval mode (BatchNorm is affine there); smaller halos are an approximation.
RECEPTIVE_HALO = 320


def scene_tiles(H, W, tile=512, halo=RECEPTIVE_HALO, strips=False):
    """Partition an (H, W) scene into ``tile`` x ``tile`` cores and their halo-extended windows, clipped to the scene (so the
    convolutions' zero padding falls on the true border, as in the whole-scene forward).  ``strips``: full-width row strips of
    ``tile`` rows with a vertical halo only -- (tile + 2 halo) / tile times the scene's work instead of ((tile + 2 halo) / tile)^2
    (2.25x instead of 5x at 512 / 320).  Returns an int64 tensor (n, 8): core y0, y1, x0, x1 and window y0, y1, x0, x1.  H, W, tile
    and halo must be multiples of 32 (encoder stride): only shifts by multiples of 32 commute with the network."""
    for v, name in ((H, "H"), (W, "W"), (tile, "tile"), (halo, "halo")):
        if v % 32 or (v <= 0 and name != "halo"):
            raise ValueError(f"scene_tiles: {name}={v} must be a positive multiple of 32")
    rows = []
    tw = W if strips else tile
    for y0 in range(0, H, tile):
        for x0 in range(0, W, tw):
            y1, x1 = min(y0 + tile, H), min(x0 + tw, W)
            rows.append((y0, y1, x0, x1, max(0, y0 - halo), min(H, y1 + halo), max(0, x0 - halo), min(W, x1 + halo)))
    return torch.tensor(rows, dtype=torch.int64)


def stitch(cores, rects, H, W):
    """cores: (n, tile, tile | W) per-tile core values (zero-padded at the right / bottom scene edge) -> (H, W)"""
    out = torch.empty((H, W), dtype=cores.dtype, device=cores.device)
    for c, (y0, y1, x0, x1) in zip(cores, rects[:, :4].tolist()):
        out[y0:y1, x0:x1] = c[:y1 - y0, :x1 - x0]
    return out


@torch.no_grad()
def tiled_logits(model, x, tile=512, halo=RECEPTIVE_HALO, batch=8, group=None, shard=None, strips=False):
    """Sliding-window inference of a (C, H, W) device scene (H, W multiples of 32): the tiles are independent work items, so with
    ``torch.distributed`` initialised they are partitioned over the ranks (``parallel.sharded_map``: no collective on the data
    path, one all_gather of the core logits at the end) -- the tile-sharded mode of BASELINE configs[4].  Windows of equal
    shape are stacked into batches of up to ``batch``.  Returns (H, W) logits on every rank.
    ``shard``: None = shard whenever a process group is initialised; False = this rank infers ALL tiles of ITS scene and no
    collective is entered (one scene per rank: ranks may hold different scenes with different tile counts)."""
    from .parallel import sharded_map
    C_, H, W = x.shape
    rects = scene_tiles(H, W, tile, halo, strips)

    def run(my):
        cores = torch.zeros((my.shape[0], tile, W if strips else tile), dtype=torch.float32, device=x.device)
        by_shape = {}
        for i, r in enumerate(my.tolist()):
            by_shape.setdefault((r[5] - r[4], r[7] - r[6]), []).append((i, r))
        for items in by_shape.values():
            for k in range(0, len(items), batch):
                chunk = items[k:k + batch]
                xb = torch.stack([x[:, r[4]:r[5], r[6]:r[7]] for _, r in chunk]).contiguous()
                lg = model(xb)
                for j, (i, r) in enumerate(chunk):
                    cores[i, :r[1] - r[0], :r[3] - r[2]] = lg[j, 0, r[0] - r[4]:r[1] - r[4], r[2] - r[6]:r[3] - r[6]]
        return cores

    cores = run(rects) if shard is False else sharded_map(run, rects, group=group)
    return stitch(cores, rects, H, W)


def _merge_column_shards(t, c0, c1, group=None):
    """every rank filtered the column blocks [c0, c1) it owns: merged by OWNERSHIP -- a rank zeroes the columns it does not own
    and the ranks SUM, so the result does not depend on how the values compare with the fill value (x + 0 + ... + 0 == x
    exactly, fill pixels inside an owned block stay fill)"""
    import torch.distributed as dist
    t[:, :c0] = 0
    t[:, c1:] = 0
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


@torch.no_grad()
def emit_scene_predict(model, raw, wavelengths, template, fill_value=-9999.0, column_step=2, num_iter=30,
                       covariance_lerp_alpha=1e-4, column_range=None, tile=None, halo=RECEPTIVE_HALO, ratio_bands=None,
                       distributed=None, group=None, strips=True):
    """``raw``: (rows, cols, S) float32 EMIT L1B radiance (device or host), ``wavelengths``: (S,) nm, ``template``: unit CH4
    absorption for the bands inside [2122, 2488] nm (``mag1c.generate_template_from_bands``; (K,) or (K, 2)).
    Returns a dict of device tensors: ``mf`` (rows, cols) ppm*m, ``albedo``, ``input`` (4, H', W') in the AVIRIS value range,
    ``prediction`` (H', W') plume probability and ``pred_binary`` (int64) -- H', W' = rows, cols cropped to multiples of 32
    (emit_dataset.py:80-93).

    ``tile``: None = one whole-scene forward (the reference's only mode); an int = sliding-window inference with ``halo``
    (:func:`tiled_logits`): full-width strips of ``tile`` rows (``strips=True``, the throughput mode: (tile + 2 halo) / tile of the
    scene's work -- 2.25x at 512 / 320 against 5x for square cores, measured 2.6x vs 6.8x wall time on 1280 x 1248) or
    ``tile`` x ``tile`` cores.  ``ratio_bands=(2350, 2310)`` additionally returns ``ratio``, the
    on-the-fly two-band ratio of feature_extration.py:42-56 on the nearest bands (absorbing, reference).
    With ``torch.distributed`` initialised (``distributed=None`` -> automatic) the column blocks of the matched filter and the
    inference tiles are partitioned over the ranks; the per-rank mf / albedo columns are merged (one all_reduce) before the
    network input is built, so every rank returns the full-scene result.  ``column_range=(c0, c1)`` (explicit manual shard,
    c0 / c1 on column_step boundaries) returns ONLY ``mf`` and ``albedo`` of that shard: a network input built from a partial
    mf would be wrong for the whole scene."""
    raw = torch.as_tensor(raw)
    dev = raw.device if raw.is_cuda else model.device
    raw = raw.to(dev).float()
    w = np.asarray(wavelengths, dtype=np.float64)
    keep = np.nonzero((w >= EMIT_MAG1C_RANGE_NM[0]) & (w <= EMIT_MAG1C_RANGE_NM[1]))[0]
    assert keep.size and np.all(np.diff(keep) == 1), "the mag1c bands must be contiguous"
    sub = raw[..., int(keep[0]):int(keep[-1]) + 1].contiguous()
    rgb = raw[..., nearest_bands(w)].permute(2, 0, 1).contiguous()
    ratio = None
    if ratio_bands is not None:
        ia, ir = nearest_bands(w, ratio_bands)
        ratio = (raw[..., ia].contiguous(), raw[..., ir].contiguous())
    return _emit_predict_parts(model, sub, rgb, ratio, template, fill_value, column_step, num_iter, covariance_lerp_alpha, column_range,
                               tile, halo, distributed, group, strips)


def _emit_predict_parts(model, sub, rgb, ratio, template, fill_value, column_step, num_iter, covariance_lerp_alpha, column_range, tile, halo,
                        distributed, group, strips=True):
    """``sub``: (rows, cols, K) device float32 radiance of the mag1c bands, ``rgb``: (3, rows, cols), ``ratio``: None or the
    (absorbing, reference) band planes -- the pieces of the cube the scene pipeline touches"""
    import torch.distributed as dist
    t = np.asarray(template, dtype=np.float64)
    t = t[:, 1] if t.ndim == 2 else t
    if t.size != sub.shape[-1]:
        raise ValueError(f"template has {t.size} bands, the cube has {sub.shape[-1]} inside {EMIT_MAG1C_RANGE_NM} nm")
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    cols = sub.shape[1]
    step = int(column_step or cols)
    if column_range is not None:
        mf, alb = mag1c.mag1c_columns(sub, t, fill_value, column_step=column_step, num_iter=num_iter,
                                      covariance_lerp_alpha=covariance_lerp_alpha, column_range=column_range)
        return {"mf": mf, "albedo": alb}
    if distributed:
        from .parallel import shard_range
        nblk = -(-cols // step)
        lo, hi = shard_range(nblk, dist.get_rank(group), dist.get_world_size(group))
        c0, c1 = lo * step, min(hi * step, cols)
        mf, alb = mag1c.mag1c_columns(sub, t, fill_value, column_step=column_step, num_iter=num_iter,
                                      covariance_lerp_alpha=covariance_lerp_alpha, column_range=(c0, c1))
        mf, alb = _merge_column_shards(mf, c0, c1, group), _merge_column_shards(alb, c0, c1, group)
    else:
        mf, alb = mag1c.mag1c_columns(sub, t, fill_value, column_step=column_step, num_iter=num_iter,
                                      covariance_lerp_alpha=covariance_lerp_alpha)
    x = emit_to_aviris_input(mf, rgb)
    out = {"mf": mf, "albedo": alb, "input": x}
    if ratio is not None:
        from .features import ratio_2c_match_c_from_sums_outlier
        out["ratio"] = ratio_2c_match_c_from_sums_outlier(ratio[0], ratio[1])
    was = model.training
    model.eval()
    try:
        logits = model(x[None]) if tile is None else tiled_logits(model, x, tile, halo, group=group, shard=bool(distributed), strips=strips)[None, None]
        masks = masks_from_logits(logits.contiguous())
    finally:
        model.train(was)
    out["prediction"], out["pred_binary"] = masks["prediction"][0, 0], masks["pred_binary"][0, 0]
    return out


@torch.no_grad()
def emit_granule_predict(model, nc_path, column_step=2, num_iter=30, covariance_lerp_alpha=1e-4, tile=None, halo=RECEPTIVE_HALO,
                         ratio_bands=None, distributed=None, group=None, rows=None, threads=8, strips=True):
    """The notebook path of the reference end to end FROM THE FILE (notebooks/inference_on_raw_EMIT_nc_file.ipynb cells 8-19:
    ``EMITImage(path)`` -> ``mag1c_emit`` -> RGB bands -> rescale -> ``model`` -> threshold): opens the EMIT L1B radiance granule
    (NetCDF-4) with :mod:`starcop_amd.hdf5_reader`, reads ONLY what the pipeline touches -- the contiguous band slice inside
    [2122, 2488] nm chunk-wise, the three RGB band planes, the two ratio bands if asked for: ~0.35 GB of the 1.8 GB cube -- builds the
    CH4 target from the file's band centres / widths (shipped look-up table) and runs :func:`emit_scene_predict`'s device pipeline.
    ``rows``: optional slice of downtrack lines.  Returns its dict plus ``wavelengths``, ``fwhm`` (the mag1c bands) and ``glt_x`` /
    ``glt_y`` (host arrays, or None) for orthorectification by the caller."""
    from .hdf5_reader import H5File
    dev = model.device
    rs = rows or slice(None)
    with H5File(nc_path) as f:
        wl = np.asarray(f["sensor_band_parameters/wavelengths"].read(), dtype=np.float64)
        fwhm = np.asarray(f["sensor_band_parameters/fwhm"].read(), dtype=np.float64)
        rad = f["radiance"]
        keep = np.nonzero((wl >= EMIT_MAG1C_RANGE_NM[0]) & (wl <= EMIT_MAG1C_RANGE_NM[1]))[0]
        assert keep.size, "There are no bands in the selected wavelength range"
        b0, b1 = int(keep[0]), int(keep[-1]) + 1
        fill = rad.attrs.get("_FillValue", rad.fillvalue)
        fill = float(fill) if fill is not None else -9999.0

        def plane(i):
            return np.ascontiguousarray(rad.read((rs, slice(None), slice(i, i + 1)), threads=threads)[..., 0], dtype=np.float32)
        sub_h = np.ascontiguousarray(rad.read((rs, slice(None), slice(b0, b1)), threads=threads), dtype=np.float32)
        rgb_h = np.stack([plane(i) for i in nearest_bands(wl)])
        ratio_h = [plane(i) for i in nearest_bands(wl, ratio_bands)] if ratio_bands is not None else None
        glt = (f["location/glt_x"].read(), f["location/glt_y"].read()) if ("location/glt_x" in f and "location/glt_y" in f) else (None, None)

    def up(a):
        h = torch.from_numpy(a)
        return (h.pin_memory() if torch.cuda.is_available() else h).to(dev, non_blocking=True)
    template = mag1c.generate_template_from_bands(wl[b0:b1], fwhm[b0:b1])
    out = _emit_predict_parts(model, up(sub_h), up(rgb_h), tuple(up(a) for a in ratio_h) if ratio_h else None, template, fill, column_step,
                              num_iter, covariance_lerp_alpha, None, tile, halo, distributed, group, strips)
    out.update(wavelengths=wl[b0:b1], fwhm=fwhm[b0:b1], glt_x=glt[0], glt_y=glt[1], fill_value=fill)
    return out


@torch.no_grad()
def aviris_scene_mag1c(aviris_img_folder, mf_filename, albedo_filename=None, use_wavelength_range=mag1c.DEFAULT_WAVELENGTH_RANGE,
                       device="cuda", extra_tags=None):
    """``run_mag1c`` of the reference (starcop/process_aviris.py:146-232) on the MI355X: opens ``<name>_img`` (radiance, ENVI)
    and ``<name>_glt`` (sample / line look-up, ENVI) of an AVIRIS-NG flight line as BIP memmaps, keeps the bands that are not
    affected by water vapour inside ``use_wavelength_range`` (they must form a slice), builds the CH4 target from the header's
    band centres and widths, runs acrwl1mf(num_iter=30) per detector sample (groups = |GLT sample index|, pixels with index 0
    are not data) and writes the result as tiled GeoTIFFs (BLOCKSIZE 128) that carry what the reference's ``save_cog`` call leaves
    in them (process_aviris.py:179-181,219-232): the radiance file's georeferencing (ENVI ``map info`` -> GeoTIFF transform + CRS
    keys), GDAL_NODATA, the band description and the dataset tags ``wavelengths`` (the kept band centres) and ``mag1c=acfwl1mf``.
    The target spectrum is cast to the radiance dtype as the reference does (:199).  Returns (mf, albedo) device tensors.
    The radiance slice is uploaded once from the memmap through pinned memory; everything after that stays on the device."""
    import os
    from . import io_formats as io
    folder = aviris_img_folder.rstrip("/")
    name = os.path.basename(folder)
    rdn, meta = io.open_envi(os.path.join(folder, f"{name}_img"))
    glt, _ = io.open_envi(os.path.join(folder, f"{name}_glt"))
    wl = meta["wavelengths"]
    keep = mag1c.get_mask_bad_bands(wl) & (wl >= use_wavelength_range[0]) & (wl <= use_wavelength_range[1])
    idx = np.flatnonzero(keep)
    assert idx[-1] - idx[0] + 1 == idx.shape[0], "Not all indexes included. Can't be a slice!"
    target = mag1c.generate_template_from_bands(centers=wl, fwhm=meta["fwhm"])
    spec = target.astype(rdn.dtype)[keep, 1]           # process_aviris.py:199: the template in the cube's dtype
    host = torch.from_numpy(np.ascontiguousarray(rdn[..., idx[0]:idx[-1] + 1], dtype=np.float32))
    x = (host.pin_memory() if torch.cuda.is_available() else host).to(device, non_blocking=True)
    groups = np.abs(np.asarray(glt[..., 0])).astype(np.int64)
    mf, alb = mag1c.func_by_groups(mag1c.Filter(spec, num_iter=30), x, groups, mask=groups != 0, max_group=int(groups.max()))
    tags = io.envi_geo_tags(meta["header"])                               # transform + crs of the radiance file
    tags[42113] = (2, (str(mag1c.NODATA),))                               # GDAL_NODATA, as fill_value_default=NODATA
    tags.update(extra_tags or {})
    md = {"wavelengths": wl[keep], "mag1c": "acfwl1mf"}
    io.write_tiff(mf_filename, mf.cpu().numpy(), blocksize=128,
                  extra_tags={**tags, **io.gdal_metadata_tag(md, ["CH4 Absorption (ppm x m)"])})
    if albedo_filename is not None:
        io.write_tiff(albedo_filename, alb.cpu().numpy(), blocksize=128, extra_tags={**tags, **io.gdal_metadata_tag(md, ["Albedo"])})
    return mf, alb
