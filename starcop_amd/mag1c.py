"""mag1c matched filters on the MI355X: same function names / arguments as the reference module
(/root/reference/starcop/models/mag1c.py: rmf :284, acrwl1mf :177, func_by_groups :117,
get_mask_bad_bands :98, generate_template_from_bands :60) plus the two drivers' group semantics
(starcop/process_aviris.py:189-219 -> :func:`acrwl1mf_by_groups`, starcop/models/mag1c_emit.py:40-90 ->
:func:`mag1c_columns`).

All groups of a scene are filtered by ONE launch of ``sc_mag1c_groups`` (a work-group per group, all 31
covariance/Cholesky rounds inside the kernel); the host only sorts pixel indices by group (torch plumbing) and
packs / scatters through ``sc_mag1c_pack`` / ``sc_scatter``.  There is no CPU fallback.
"""
import ctypes as C
import os
import re

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, sc_mag1c_args, stream

NODATA = -9999
SCALING = 1e5
EPSILON = 1e-9
MAX_BANDS = 128
DEFAULT_WAVELENGTH_RANGE = (2122, 2488)


# ------------------------------------------------------------------------------------------------
def _run_groups(cube2d, S_total, band0, S, pix_index, counts, template, num_iter, alpha, k, flags, statmask=None, energy=False):
    """cube2d: (npixels_image, S_total) device tensor (f32|f64, contiguous); pix_index: int64 device tensor of the packed
    pixels (group after group); counts: python list / cpu tensor of pixels per group.  Returns (mf, albedo) packed."""
    lib = _lib.load()
    dev = cube2d.device
    is64 = cube2d.dtype == torch.float64
    counts_t = torch.as_tensor(counts, dtype=torch.int64)
    G = int(counts_t.numel())
    npix = int(counts_t.sum())
    dt = cube2d.dtype
    mf = torch.empty(npix, dtype=dt, device=dev)
    alb = torch.empty(npix, dtype=dt, device=dev)
    if G == 0 or npix == 0:
        return mf, alb
    if S > MAX_BANDS:
        raise ValueError(f"mag1c: at most {MAX_BANDS} bands per filter (got {S})")
    ppad = ((counts_t + 63) // 64) * 64
    poff = torch.cumsum(counts_t, 0) - counts_t
    xoff = torch.cumsum(ppad * S, 0) - ppad * S
    total = int((ppad * S).sum())
    xp = torch.empty(total, dtype=dt, device=dev)                # sc_mag1c_pack writes the padding of every band row too
    P_d = counts_t.to(torch.int32).to(dev)
    ppad_d = ppad.to(torch.int32).to(dev)
    poff_d, xoff_d = poff.to(dev), xoff.to(dev)
    st = stream()
    check(lib.sc_mag1c_pack(ptr(cube2d), 1 if is64 else 0, S_total, band0, S, ptr(pix_index), ptr(xoff_d), ptr(ppad_d),
                            ptr(poff_d), ptr(P_d), G, ptr(xp), 1 if is64 else 0, st))
    a = sc_mag1c_args()
    a.x = xp.data_ptr(); a.x_is_f64 = 1 if is64 else 0
    a.xoff = xoff_d.data_ptr(); a.P = P_d.data_ptr(); a.Ppad = ppad_d.data_ptr(); a.poff = poff_d.data_ptr()
    a.statmask = statmask.data_ptr() if statmask is not None else None
    a.G, a.S, a.npix = G, S, npix
    templ = _template_on(dev, template, S)
    a.templ = templ.data_ptr()
    a.num_iter, a.alpha, a.cov_update_scaling = int(num_iter), float(alpha), float(k)
    a.albedo_override, a.zero_override, a.sparse_override, a.apply_scaling = (int(bool(f)) for f in flags)
    work = torch.empty(lib.sc_mag1c_workspace_doubles(G, S, npix), dtype=torch.float64, device=dev)
    status = torch.zeros(G, dtype=torch.int32, device=dev)
    a.work, a.mf_out, a.albedo_out, a.status = work.data_ptr(), mf.data_ptr(), alb.data_ptr(), status.data_ptr()
    en = ld = None
    if energy:          # compute_energy: per group and stage the residual term, per group the log-determinant term of the rmf stage
        en = torch.zeros(G, max(int(num_iter), 0) + 1, dtype=torch.float64, device=dev)
        ld = torch.zeros(G, dtype=torch.float64, device=dev)
        a.energy, a.logdet = en.data_ptr(), ld.data_ptr()
    check(lib.sc_mag1c_groups(C.byref(a), st))
    bad = torch.nonzero(status).reshape(-1)
    if bad.numel():          # the reference's torch.linalg.cholesky raises (mag1c.py:251,323)
        raise torch.linalg.LinAlgError(
            f"linalg.cholesky: (Batch element {int(bad[0])}): The factorization could not be completed because the "
            "input is not positive-definite")
    return (mf, alb, en, ld) if energy else (mf, alb)


def _column_runs(ids):
    """ids: 1-D host int array (group id of every image column).  -> (gcol int32 [G+1], sorted-by-column) if every id
    occupies ONE contiguous run of columns, else None."""
    ids = np.asarray(ids)
    cuts = np.flatnonzero(np.diff(ids) != 0) + 1
    starts = np.concatenate([[0], cuts])
    if np.unique(ids[starts]).size != starts.size:
        return None
    return np.concatenate([starts, [ids.size]]).astype(np.int32)


_LAYOUT_CACHE = {}
MAX_SORT_WORK_INTS = 64 << 20        # histogram ints of the device counting sort (256 MB); beyond: the general sort path
MAX_GROUP_IDS = 1 << 16      # group ids up to this take the device counting sort (detector widths are ~600-1300); larger / negative ids
                             # fall back to the general sort
COLUMN_FAST_PATH = True      # False: always take the general (device sort) layout path; tests compare the two bit for bit
DIRECT_TILES = os.environ.get("STARCOP_MAG1C_DIRECT", "1") == "1"    # column groups of <= 512 pixels (a 512-row tile per detector column):
                             # the filter kernel gathers from the cube and scatters to image order itself (no pack / scatter passes)
_TEMPLATE_CACHE = {}


def _template_on(dev, template, S):
    """the unit absorption spectrum as a float64 device tensor; the upload is cached by content (drivers pass the same array per tile)"""
    t_host = template.detach().cpu().numpy() if torch.is_tensor(template) else np.asarray(template)
    t_host = np.ascontiguousarray(t_host, dtype=np.float64).reshape(-1)
    if t_host.size != S:
        raise ValueError(f"mag1c: template has {t_host.size} bands, data has {S}")
    key = (str(dev), t_host.tobytes())
    t = _TEMPLATE_CACHE.get(key)
    if t is None:
        if len(_TEMPLATE_CACHE) > 16:
            _TEMPLATE_CACHE.clear()
        t = _TEMPLATE_CACHE[key] = torch.from_numpy(t_host.copy()).to(dev)
    return t


def _run_column_groups(cube3, b0, S, valid_u8, gcol, min_keep, template, num_iter, alpha, k, flags, fill, out_dtype, ids=None, nids=0):
    """Layout, pack, filter and scatter entirely on the device -- ``sc_mag1c_layout_columns`` (column-structured groups: both
    drivers of the reference on un-orthorectified cubes) or ``sc_mag1c_layout_ids`` (``ids``: int32 device tensor of a group id
    per pixel in [0, nids): the orthorectified |GLT sample| case) -> ``sc_mag1c_pack`` -> ``sc_mag1c_groups`` -> ``sc_scatter_n``
    with every per-group array and the pixel count in device memory; the only host synchronisation is the status check after
    the filter."""
    lib = _lib.load()
    dev = cube3.device
    rows, cols, S_total = cube3.shape
    HW = rows * cols
    G = int(gcol.size - 1) if ids is None else int(nids)
    is64 = cube3.dtype == torch.float64
    dt = cube3.dtype
    both = torch.full((2, HW), fill, dtype=out_dtype, device=dev)           # (one fill launch for the two products)
    mf_out, alb_out = both[0], both[1]
    if G == 0:
        return mf_out.reshape(rows, cols), alb_out.reshape(rows, cols)
    if S > MAX_BANDS:
        raise ValueError(f"mag1c: at most {MAX_BANDS} bands per filter (got {S})")
    gcol_d = None
    if ids is None:
        key = (dev, gcol.tobytes())
        gcol_d = _LAYOUT_CACHE.get(key)
        if gcol_d is None:
            if len(_LAYOUT_CACHE) > 16:
                _LAYOUT_CACHE.clear()
            gcol_d = _LAYOUT_CACHE[key] = torch.from_numpy(gcol).to(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    # the per-group arrays and the pixel list from ONE allocation (every torch.empty is ~5 us of host time in a 0.9 ms call)
    ibuf = torch.empty(HW + 3 * G + 2, dtype=torch.int64, device=dev)
    pix, poff_d, xoff_d, totals = ibuf[:HW], ibuf[HW:HW + G], ibuf[HW + G:HW + 2 * G], ibuf[HW + 2 * G:HW + 2 * G + 2]
    pp = ibuf[HW + 2 * G + 2:].view(torch.int32)
    P_d, ppad_d = pp[:G], pp[G:2 * G]
    st = stream()
    if ids is None:
        check(lib.sc_mag1c_layout_columns(ptr(valid_u8), rows, cols, ptr(gcol_d), G, S, int(min_keep), ptr(P_d), ptr(ppad_d),
                                          ptr(poff_d), ptr(xoff_d), ptr(pix), ptr(totals), st))
    else:           # arbitrary integer groups (orthorectified cube): stable counting sort on the device
        work = torch.empty(lib.sc_mag1c_layout_ids_workspace_ints(HW, G), **i32)
        check(lib.sc_mag1c_layout_ids(ptr(valid_u8), ptr(ids), HW, G, S, int(min_keep), ptr(P_d), ptr(ppad_d), ptr(poff_d), ptr(xoff_d),
                                      ptr(pix), ptr(totals), ptr(work), st))
    # DIRECT: fp32 cube, 65..128 bands and no group that can exceed the 512 pixels the filter kernel keeps in registers (column groups:
    # rows x widest run of columns) -- the kernel reads its pixels from the cube through `pix` and writes image order itself
    # Arbitrary integer groups (the orthorectified |GLT sample| map of the AVIRIS-NG driver): their sizes are known on the device only, so
    # the direct launch is TRIED when the mean group is small enough (a 512^2 tile of a 598-sample detector: ~438 pixels per group); a
    # group beyond 512 pixels reports status 2 and the call is redone on the packed path (whole flight lines never try: HW / G > 512).
    direct = (DIRECT_TILES and not is64 and S > 64 and out_dtype in (torch.float32, torch.float64)
              and (rows * int(np.diff(gcol).max()) <= 512 if ids is None else HW <= 512 * G))
    a = sc_mag1c_args()
    a.x_is_f64 = 1 if is64 else 0
    a.P = P_d.data_ptr(); a.poff = poff_d.data_ptr()
    a.statmask = None
    a.G, a.S, a.npix = G, S, HW
    templ = _template_on(dev, template, S)
    a.templ = templ.data_ptr()
    a.num_iter, a.alpha, a.cov_update_scaling = int(num_iter), float(alpha), float(k)
    a.albedo_override, a.zero_override, a.sparse_override, a.apply_scaling = (int(bool(f)) for f in flags)
    work = torch.empty(lib.sc_mag1c_workspace_doubles(G, S, HW), dtype=torch.float64, device=dev)
    status = torch.zeros(G, dtype=torch.int32, device=dev)
    a.work, a.status = work.data_ptr(), status.data_ptr()
    o64 = 1 if out_dtype == torch.float64 else 0
    if direct:
        a.cube, a.S_total, a.band0, a.pix_index = cube3.data_ptr(), S_total, b0, pix.data_ptr()
        a.scatter_mf, a.scatter_alb, a.scatter_is_f64 = mf_out.data_ptr(), alb_out.data_ptr(), o64
        check(lib.sc_mag1c_groups(C.byref(a), st))
        if ids is not None and int(status.max()) == 2:      # a group of more than 512 pixels: the whole call again, packed
            both.fill_(fill)
            status.zero_()
            a.cube = None
            direct = False
    if not direct:
        xp = torch.empty((HW + 64 * G) * S, dtype=dt, device=dev)            # upper bound of sum(Ppad) * S: no size read-back, no memset
        check(lib.sc_mag1c_pack(ptr(cube3), 1 if is64 else 0, S_total, b0, S, ptr(pix), ptr(xoff_d), ptr(ppad_d),
                                ptr(poff_d), ptr(P_d), G, ptr(xp), 1 if is64 else 0, st))
        a.x = xp.data_ptr()
        a.xoff = xoff_d.data_ptr(); a.Ppad = ppad_d.data_ptr()
        mf, alb = torch.empty(HW, dtype=dt, device=dev), torch.empty(HW, dtype=dt, device=dev)
        a.mf_out, a.albedo_out = mf.data_ptr(), alb.data_ptr()
        check(lib.sc_mag1c_groups(C.byref(a), st))
        check(lib.sc_scatter_n(ptr(mf), 1 if is64 else 0, ptr(pix), ptr(totals), HW, ptr(mf_out), o64, st))
        check(lib.sc_scatter_n(ptr(alb), 1 if is64 else 0, ptr(pix), ptr(totals), HW, ptr(alb_out), o64, st))
    if int(status.max()):          # the reference's torch.linalg.cholesky raises (mag1c.py:251,323)
        if int((status == 2).sum()):          # (cannot happen: `direct` bounds every group by rows x widest run)
            raise RuntimeError("mag1c: a group exceeded the 512 pixels of the direct tile path")
        bad = torch.nonzero(status).reshape(-1)
        raise torch.linalg.LinAlgError(
            f"linalg.cholesky: (Batch element {int(bad[0])}): The factorization could not be completed because the "
            "input is not positive-definite")
    return mf_out.reshape(rows, cols), alb_out.reshape(rows, cols)


def _batched(x, template, num_iter, alpha, k, flags, mask, energy=False):
    _lib.require_device(x)
    if x.dim() != 3:
        raise ValueError("x must be [batch, pixels, spectrum]")
    if x.dtype not in (torch.float32, torch.float64):
        x = x.float()
    b, p, s = x.shape
    x2 = x.contiguous().reshape(b * p, s)
    pix = torch.arange(b * p, dtype=torch.int64, device=x.device)
    sm = None
    if mask is not None:
        m = torch.as_tensor(mask).to(x.device)
        m = torch.squeeze(m, 0) if m.dim() > 1 else m
        assert m.shape == x.shape[1:2], f"Unexpected shape of mask: {m.shape} expected {x.shape[1:2]}"
        sm = m.to(torch.uint8).repeat(b).contiguous()
    out = _run_groups(x2, s, 0, s, pix, [p] * b, template, num_iter, alpha, k, flags, sm, energy=energy)
    return (out[0].reshape(b, p, 1), out[1].reshape(b, p, 1)) + tuple(out[2:])


@torch.no_grad()
def rmf(x, template, alpha=0., zero_override=False, compute_energy=False, albedo_override=False, apply_scaling=True,
        mask=None):
    """Classic robust matched filter, [b, p, s] -> (mf [b, p, 1] ppm*m, albedo [b, p, 1]).
    ``compute_energy`` (mag1c.py:337-343): also the reference's scalar -- the residual term summed over the batch + N/2 log(1 / prod of the
    diagonals of all Cholesky factors) -- and, as there, mf is returned WITHOUT the ppm*m scaling (the reference returns before that line).
    The residual term is the sum of all entries of (x-mu) C^{-1} (x-mu)^T = s^T C^{-1} s, s = sum_p (x_p - mu): evaluated in that form
    (fp64 arithmetic), not as a P x P matrix; without a mask it is exactly zero where the reference returns rounding noise."""
    if compute_energy:
        mf, alb, en, ld = _batched(x, template, -1, alpha, 1.0, (albedo_override, zero_override, False, False), mask, energy=True)
        return mf, alb, (en[:, 0].sum() + ld.sum()).to(mf.dtype)
    return _batched(x, template, -1, alpha, 1.0, (albedo_override, zero_override, False, apply_scaling), mask)


@torch.no_grad()
def acrwl1mf(x, template, num_iter=30, albedo_override=False, zero_override=False, sparse_override=False,
             covariance_update_scaling=1., alpha=0., compute_energy=False, mask=None):
    """Albedo-corrected reweighted-L1 matched filter, [b, p, s] -> (mf [b, p, 1], albedo [b, p, 1]).
    ``compute_energy`` (mag1c.py:270-275): also the list the reference returns -- [the rmf energy, then one residual term per iteration,
    each summed over the batch] (see rmf for how the terms are evaluated)."""
    if compute_energy:
        mf, alb, en, ld = _batched(x, template, int(num_iter), alpha, covariance_update_scaling,
                                   (albedo_override, zero_override, sparse_override, True), mask, energy=True)
        tot = en.sum(dim=0).to(mf.dtype)
        return mf, alb, [tot[0] + ld.sum().to(mf.dtype)] + [tot[1 + i] for i in range(int(num_iter))]
    return _batched(x, template, int(num_iter), alpha, covariance_update_scaling,
                    (albedo_override, zero_override, sparse_override, True), mask)


class Filter:
    """Parameters of the per-group filter; what the reference passes as ``lambda x: acrwl1mf(x, spec, num_iter=30)``."""

    def __init__(self, template, num_iter=30, alpha=0., albedo_override=False, zero_override=False, sparse_override=False,
                 covariance_update_scaling=1.):
        self.template, self.num_iter, self.alpha = template, num_iter, alpha
        self.flags = (albedo_override, zero_override, sparse_override, True)
        self.k = covariance_update_scaling

    def __call__(self, x):
        return acrwl1mf(x, self.template, num_iter=self.num_iter, alpha=self.alpha, albedo_override=self.flags[0],
                        zero_override=self.flags[1], sparse_override=self.flags[2], covariance_update_scaling=self.k)


@torch.no_grad()
def func_by_groups(func, x, groups, mask=None, disable_pbar=True, samples_read=50, band_slice=None, max_group=None):
    """(H, W, S) radiance + (H, W) integer groups -> (mf, albedo) (H, W) tensors, NODATA where not computed.

    Every group id present under ``mask`` is filtered on its own valid pixels; groups with <= 10 valid pixels are
    skipped (mag1c.py:166).  ``func`` should be a :class:`Filter` (all groups in one kernel launch); any other callable
    is applied group by group as the reference does."""
    x = torch.as_tensor(x)
    _lib.require_device(x if x.is_cuda else None)
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = x.to(dev)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.float()
    H, W, S_total = x.shape
    b0, b1 = (0, S_total) if band_slice is None else (band_slice.start or 0, band_slice.stop or S_total)
    groups_dev = None

    def groups_on_device():          # (only the non-column paths need the group map on the device: 2 MB per 512^2 tile)
        nonlocal groups_dev
        if groups_dev is None:
            groups_dev = (groups if torch.is_tensor(groups) else torch.as_tensor(np.asarray(groups))).to(dev).reshape(-1).long()
        return groups_dev
    mask_t = None
    if mask is None:
        xc = x.contiguous()
        mask_u8 = torch.empty(H * W, dtype=torch.uint8, device=dev)
        check(_lib.load().sc_valid_mask(ptr(xc), 1 if xc.dtype == torch.float64 else 0, S_total, b0, b1 - b0, float(NODATA),
                                        H * W, ptr(mask_u8), stream()))
    else:
        mask_t = (mask if torch.is_tensor(mask) else torch.as_tensor(np.asarray(mask))).to(dev).reshape(-1).bool()
    if isinstance(func, Filter) and COLUMN_FAST_PATH:
        # both drivers of the reference group by detector column(s): when every row of `groups` is the same and each id is one
        # run of columns, the layout is built on the device without the sort below (results are the same pixels in the same order)
        g_host = groups.cpu().numpy() if torch.is_tensor(groups) else np.asarray(groups)
        g_host = g_host.reshape(H, W)
        gcol = _column_runs(g_host[0]) if bool((g_host == g_host[:1]).all()) else None
        if gcol is not None:
            xc = x.contiguous()
            return _run_column_groups(xc, b0, b1 - b0, mask_t.to(torch.uint8).contiguous() if mask is not None else mask_u8, gcol, 10,
                                      func.template, func.num_iter, func.alpha, func.k, func.flags, NODATA, x.dtype)
    if isinstance(func, Filter) and COLUMN_FAST_PATH:
        # any other integer group map (the orthorectified GLT grid of process_aviris.py:211-217): counting sort on the device.
        # The ids are bounded by the detector width; max_group (or one .max() read-back) sizes the histogram.
        # max_group is a caller's promise: ids above it (or negative ones) would be dropped to NODATA without a word, so the data
        # are checked against it (one read-back either way)
        groups_t = groups_on_device()
        gmin, gmax = (int(v) for v in torch.aminmax(groups_t))          # one read-back for both
        if max_group is not None:
            if gmax > int(max_group) or gmin < 0:
                raise ValueError(f"func_by_groups: group ids span [{gmin}, {gmax}] but max_group={max_group} was given")
            gmax = int(max_group)
        # the counting sort keeps ceil(HW/1024) * nids ints of histograms and pads every id's pixels to 64: fine for ids bounded by
        # a detector width, ruinous for sparse label ids (60 000 ids on a 30 Mpx flight line = 7.6 GB): those take the general path
        nblk = -(-(H * W) // 1024)
        if gmin >= 0 and gmax < MAX_GROUP_IDS and nblk * (gmax + 1) <= MAX_SORT_WORK_INTS:
            xc = x.contiguous()
            return _run_column_groups(xc, b0, b1 - b0, mask_t.to(torch.uint8).contiguous() if mask is not None else mask_u8, None, 10,
                                      func.template, func.num_iter, func.alpha, func.k, func.flags, NODATA, x.dtype,
                                      ids=groups_t.to(torch.int32).contiguous(), nids=gmax + 1)
    groups_t = groups_on_device()
    if mask_t is None:
        mask_t = mask_u8.bool()
    mf_out = torch.full((H * W,), NODATA, dtype=x.dtype, device=dev)
    alb_out = torch.full((H * W,), NODATA, dtype=x.dtype, device=dev)
    valid_idx = torch.nonzero(mask_t).reshape(-1)
    if valid_idx.numel() == 0:
        return mf_out.reshape(H, W), alb_out.reshape(H, W)
    gv = groups_t[valid_idx]
    order = torch.argsort(gv, stable=True)
    pix_sorted = valid_idx[order]
    uniq, counts = torch.unique_consecutive(gv[order], return_counts=True)
    keep = counts > 10
    starts = torch.cumsum(counts, 0) - counts
    if isinstance(func, Filter):
        sel = torch.repeat_interleave(keep, counts)
        pix = pix_sorted[sel].contiguous()
        cnt = counts[keep].cpu()
        mf, alb = _run_groups(x.reshape(H * W, S_total).contiguous(), S_total, b0, b1 - b0, pix, cnt, func.template,
                              func.num_iter, func.alpha, func.k, func.flags)
        lib = _lib.load()
        is64 = 1 if x.dtype == torch.float64 else 0
        check(lib.sc_scatter(ptr(mf), is64, ptr(pix), pix.numel(), ptr(mf_out), is64, stream()))
        check(lib.sc_scatter(ptr(alb), is64, ptr(pix), pix.numel(), ptr(alb_out), is64, stream()))
    else:
        xf = x.reshape(H * W, S_total)
        for s0, c, k in zip(starts.tolist(), counts.tolist(), keep.tolist()):
            if not k:
                continue
            pix = pix_sorted[s0:s0 + c]
            mf, alb = func(xf[pix][:, b0:b1].unsqueeze(0))
            mf_out[pix], alb_out[pix] = mf[0, :, 0].to(x.dtype), alb[0, :, 0].to(x.dtype)
    return mf_out.reshape(H, W), alb_out.reshape(H, W)


@torch.no_grad()
def acrwl1mf_by_groups(x, template, groups, mask=None, num_iter=30, alpha=0., band_slice=None, max_group=None):
    """AVIRIS-NG driver core (process_aviris.py:209-219): acrwl1mf(num_iter=30, alpha=0) per detector column.  ``max_group``: an
    upper bound of the group ids if the caller knows it (the detector width): saves the one device read-back that sizes the
    counting sort of a non-column-structured (orthorectified) group map."""
    return func_by_groups(Filter(template, num_iter=num_iter, alpha=alpha), x, groups, mask, band_slice=band_slice, max_group=max_group)


@torch.no_grad()
def mag1c_columns(raw, template, fill_value=-9999.0, column_step=None, num_iter=30, covariance_lerp_alpha=1e-4,
                  column_range=None):
    """EMIT driver core (mag1c_emit.py:50-90): raw (rows, cols, S) float32 radiance; blocks of ``column_step`` columns
    (None: whole image) are filtered independently in float64 on their valid pixels (no band equal to ``fill_value``);
    returns float32 (rows, cols) mf and albedo filled with ``fill_value``.  ``column_range=(c0, c1)`` restricts the work
    to a shard of column blocks (multi-GPU: groups are independent, no collective)."""
    raw = torch.as_tensor(raw)
    dev = raw.device if raw.is_cuda else torch.device("cuda", torch.cuda.current_device())
    _lib.require_device()
    raw = raw.to(dev).float().contiguous()
    rows, cols, S = raw.shape
    step = int(column_step or cols)
    lib = _lib.load()
    # pixels with any band equal to the fill value are left out (mag1c_emit.py:60-66): one pass over the cube
    valid = torch.empty(rows * cols, dtype=torch.uint8, device=dev)
    check(lib.sc_valid_mask_ne(ptr(raw), 0, S, 0, S, float(fill_value), rows * cols, ptr(valid), stream()))
    if column_range is not None:
        c0, c1 = int(column_range[0]), int(column_range[1])
        if c0 % step or (c1 % step and c1 != cols):
            raise ValueError(f"mag1c_columns: column_range {column_range} must fall on column_step={step} boundaries: a "
                             "truncated block would be filtered with different statistics than in the whole scene")
        edges = np.arange(c0, c1, step)
        gcol = np.concatenate([edges, [c1]]).astype(np.int32)
    else:
        gcol = np.concatenate([np.arange(0, cols, step), [cols]]).astype(np.int32)
    # The reference converts the float32 radiances to float64 before filtering (:74-75).  The kernels evaluate every
    # statistic, factorisation and per-pixel product in fp64 whatever the storage type, and double(float32) is exact, so
    # the float32 cube is filtered as it is: the same arithmetic on the same values at half the HBM traffic (the EMIT
    # path streams X 62 times and is bandwidth-bound); the results are rounded to float32 once, as the reference does (:90).
    # Pixel order inside a block follows the reference's boolean indexing raw[:, c0:c1][valid] (row-major).
    return _run_column_groups(raw, 0, S, valid, gcol, 0, template, num_iter, covariance_lerp_alpha, 1.0,
                              (False, False, False, True), float(fill_value), torch.float32)


# ------------------------------------------------------------------------------------------------
def get_mask_bad_bands(wave):
    """Bands to keep: 400..2485 nm without the water-vapour windows (1350,1420) and (1800,1945) nm."""
    w = np.asarray(wave)
    reject = (w < 400) | (w > 2485) | ((w > 1350) & (w < 1420)) | ((w > 1800) & (w < 1945))
    return ~reject


def _lut_paths(lut_dir=None):
    cands = [lut_dir, os.environ.get("STARCOP_CH4_LUT_DIR"), os.path.join(os.path.dirname(__file__), "data")]
    for d in cands:
        if d and os.path.exists(os.path.join(d, "ch4.lut")) and os.path.exists(os.path.join(d, "ch4.hdr")):
            return os.path.join(d, "ch4.hdr"), os.path.join(d, "ch4.lut")
    raise FileNotFoundError(
        "CH4 look-up table (ch4.hdr + ch4.lut, mag1c upstream, 1.8 MB) not found: pass lut_dir=... or set "
        "STARCOP_CH4_LUT_DIR to the directory that holds them (starcop/models/ in the reference checkout)")


def read_ch4_lut(lut_dir=None):
    """ENVI BSQ float64 reader -> (radiance [7, n_wave], wavelengths [n_wave] nm)."""
    hdr, dat = _lut_paths(lut_dir)
    txt = open(hdr).read()
    dims = {k: int(re.search(rf"{k}\s*=\s*(\d+)", txt).group(1)) for k in ("samples", "lines", "bands")}
    wl = re.search(r"wavelength\s*=\s*\{([^}]*)\}", txt, re.S).group(1)
    wave = np.array([float(v) for v in wl.replace("\n", " ").split(",") if v.strip()])
    arr = np.fromfile(dat, dtype="<f8").reshape(dims["bands"], dims["lines"], dims["samples"])
    return arr.transpose(1, 2, 0).squeeze(), wave


def generate_template_from_bands(centers, fwhm, lut_dir=None):
    """Unit CH4 absorption spectrum for a sensor's band set -> (K, 2) [center, spectrum] (one-off, host, fp64).
    Bands that do not overlap the LUT's 1400-2522 nm span come back as NaN (the reference leaves them undefined)."""
    centers, fwhm = np.asarray(centers, dtype=np.float64), np.asarray(fwhm, dtype=np.float64)
    if np.any(~np.isfinite(centers)) or np.any(~np.isfinite(fwhm)):
        raise RuntimeError("Band Wavelengths Centers/FWHM data contains non-finite data (NaN or Inf).")
    if centers.shape[0] != fwhm.shape[0]:
        raise RuntimeError("Length of band center wavelengths and band fwhm arrays must be equal.")
    rads, wave = read_ch4_lut(lut_dir)
    conc = np.array([0., 500., 1000., 2000., 4000., 8000., 16000.])
    sigma2 = (fwhm / (2.0 * np.sqrt(2.0 * np.log(2.0)))) ** 2
    resp = np.exp(-(wave[:, None] - centers[None, :]) ** 2 / (2 * sigma2)) / np.sqrt(2 * np.pi * sigma2)
    mass = resp.sum(axis=0)
    spectrum = np.full(centers.shape[0], np.nan)
    inside = mass > 0
    rs = rads @ (resp[:, inside] / mass[inside])
    pos = np.all(rs > 0, axis=0)
    design = np.stack([np.ones_like(conc), conc], axis=1)
    coef = np.linalg.lstsq(design, np.log(rs[:, pos]), rcond=None)[0]
    spectrum[np.flatnonzero(inside)[pos]] = coef[1] * SCALING
    return np.stack([centers, spectrum], axis=1)
