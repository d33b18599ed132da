"""Thin test-side helpers that call the C ABI (include/starcop_hip.h) for single ops."""
import ctypes as C

import torch

from starcop_amd import _lib
from starcop_amd._lib import (ACT_NONE, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_NORM, SRC_RAW, STAT_CONV1, STAT_CONV1K, STAT_CONV3, check,
                              make_src, ptr, sc_conv_args, sc_wgrad_args, stream)

DEV = "cuda"
_KEEP = []      # device tensors referenced only through raw pointers must outlive the launch


def dev(t):
    """host tensor -> device tensor that stays alive until the end of the test"""
    d = t.detach().to(DEV).contiguous()
    _KEEP.append(d)
    return d


def cst_affine(scale, shift):
    c = torch.zeros(scale.numel(), SC_CST, device=DEV)
    c[:, 0], c[:, 1] = scale.to(DEV), shift.to(DEV)
    _KEEP.append(c)
    return c


def pack(w, co_t, tflip):
    lib = _lib.load()
    co, ci, ks = w.shape[0], w.shape[1], w.shape[2]
    out = torch.empty(lib.sc_packed_weight_floats(co, ci, ks, co_t, tflip), device=DEV)
    check(lib.sc_pack_weights(ptr(w), ptr(out), co, ci, ks, co_t, tflip, stream()))
    return out


# the split the un-parametrised bx3 tests run with: 3 = three bf16 terms, 4 = two fp16 terms (SC_TERMS_F16X2, the network default);
# tests/test_gpu_ops.py and tests/test_gpu_fuzz.py run their split-kernel cases under both (fixture `split_mode`)
DEFAULT_BX3_TERMS = 3


def pack_bx3(w, co_t, tflip, terms=None):
    terms = DEFAULT_BX3_TERMS if terms is None else terms
    lib = _lib.load()
    co, ci = w.shape[0], w.shape[1]
    out = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, co_t, tflip, terms), device=DEV)
    check(lib.sc_pack_weights_bx3(ptr(w), ptr(out), co, ci, co_t, tflip, terms, stream()))
    return out


def conv_mfma(srcs, wpk, N, H, W, Cout, ks, co_t, want_stats=False, csplit=None, add0=None, add1=None,
              accum=None, outs=None, bx3=False, ksplit=False, terms=0, down0=False, absmax=None, bnr=None):
    """bnr = (y, cst, act) of out0's tensor: the launch also leaves its BatchNorm-backward partial rows and range hint -> LAST_BNR"""
    lib = _lib.load()
    a = sc_conv_args()
    a.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, Cout, ks, co_t
    csplit = Cout if csplit is None else csplit
    if outs is None:
        outs = [torch.empty(N, csplit, H // 2, W // 2, device=DEV) if down0 else torch.empty(N, csplit, H, W, device=DEV)]
        if csplit < Cout:
            outs.append(torch.empty(N, Cout - csplit, H, W, device=DEV))
    a.out0 = outs[0].data_ptr()
    a.out1 = outs[1].data_ptr() if len(outs) > 1 else None
    a.csplit = csplit
    a.terms = terms if (terms or not bx3) else DEFAULT_BX3_TERMS
    a.down0 = 1 if down0 else 0
    a.absmax = absmax.data_ptr() if absmax is not None else None
    a.accum0, a.accum1 = (accum or (0, 0))
    a.add0 = add0.data_ptr() if add0 is not None else None
    a.add1 = add1.data_ptr() if add1 is not None else None
    rows = lib.sc_stat_rows(STAT_CONV3 if ks == 3 else (STAT_CONV1K if ksplit else STAT_CONV1), N, H, W)
    stats = torch.full((rows, Cout, 2), float("nan"), device=DEV) if want_stats else None      # every entry must be written
    a.stats = stats.data_ptr() if want_stats else None
    if bnr is not None:
        a.bnr = C.addressof(make_bnr(bnr, rows, csplit))
    fn = lib.sc_conv3x3_bx3 if bx3 else (lib.sc_conv1x1_ksplit if ksplit else lib.sc_conv2d_mfma)
    check(fn(C.byref(a), stream()))
    return outs, stats


LAST_BNR = None


def make_bnr(bnr, rows, C_):
    """sc_bnr_args for (y, cst, act); the rows (NaN-filled: every entry must be written) and the zeroed range slot -> LAST_BNR"""
    global LAST_BNR
    from starcop_amd._lib import sc_bnr_args
    y, cst, act = bnr
    b = sc_bnr_args()
    r = torch.full((rows, C_, 2), float("nan"), device=DEV)
    amax = torch.zeros(1, device=DEV)
    b.y, b.cst, b.act, b.rows, b.absmax = y.data_ptr(), cst.data_ptr(), act, r.data_ptr(), amax.data_ptr()
    _KEEP.extend([b, y, cst])
    LAST_BNR = (r, amax)
    return b


def wgrad_mfma(dy, srcs, N, H, W, Cout, Cin, ks, bx3=False, terms=0, absmax=None):
    lib = _lib.load()
    a = sc_wgrad_args()
    a.dy = dy
    a.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, Cout, Cin, ks
    a.terms = terms if (terms or not bx3) else DEFAULT_BX3_TERMS
    a.absmax = absmax.data_ptr() if absmax is not None else None
    n = lib.sc_wgrad_bx3_workspace_floats(N, H, W, Cout, Cin) if bx3 else lib.sc_wgrad_workspace_floats(N, H, W, Cout, Cin, ks)
    ws = torch.empty(n, device=DEV)
    a.part, a.part_floats = ws.data_ptr(), n
    dw = torch.empty(Cout, Cin, ks, ks, device=DEV)
    a.dw = dw.data_ptr()
    check((lib.sc_conv3x3_wgrad_bx3 if bx3 else lib.sc_conv2d_wgrad_mfma)(C.byref(a), stream()))
    return dw


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-3))


def pack_pw3(w, tflip):
    """pointwise filter (Cout, Cin, 1, 1) -> the sc_conv1x1_pw3 layout through the batched pack launch (the network's path)"""
    import numpy as np
    from starcop_amd._lib import PACK_PW3
    lib = _lib.load()
    co, ci = w.shape[0], w.shape[1]
    out = torch.zeros(lib.sc_packed_weight_floats_pw3(co, ci, tflip), device=DEV)
    total = lib.sc_pack_work_items(co, ci, 1, 0, tflip, PACK_PW3)
    dt = np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("co_t", "<i4"),
                   ("tflip", "<i4"), ("bx3", "<i4"), ("total", "<u8")])
    descs = torch.from_numpy(np.array([(w.data_ptr(), out.data_ptr(), co, ci, 1, 0, tflip, PACK_PW3, total)], dtype=dt).view(np.uint8).copy()).to(DEV)
    starts = torch.zeros(1, dtype=torch.int32, device=DEV)
    _KEEP.extend([descs, starts])
    check(lib.sc_pack_weights_batch(ptr(descs), ptr(starts), 1, -(-total // 256), stream()))
    return out


def conv_pw3(src, wpk, N, H, W, Cout, want_stats=False, add0=None, accum_into=None, bnr=None):
    from starcop_amd._lib import STAT_PW3
    lib = _lib.load()
    a = sc_conv_args()
    a.nsrc = 1
    a.src[0] = src
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, Cout, 1, 32
    out = accum_into if accum_into is not None else torch.full((N, Cout, H, W), float("nan"), device=DEV)
    a.out0, a.out1, a.csplit = out.data_ptr(), None, Cout
    a.accum0 = 1 if accum_into is not None else 0
    a.add0 = add0.data_ptr() if add0 is not None else None
    rows = lib.sc_stat_rows(STAT_PW3, N, H, W)
    stats = torch.full((rows, Cout, 2), float("nan"), device=DEV) if want_stats else None
    a.stats = stats.data_ptr() if want_stats else None
    if bnr is not None:      # (y, cst, act) of out0's tensor: the launch leaves its BatchNorm-backward rows (one per 32-pixel block) -> LAST_BNR
        b = make_bnr(bnr, -(-(N * H * W) // 32), Cout)
        b.absmax = None      # (no range hint on this path)
        a.bnr = C.addressof(b)
    check(lib.sc_conv1x1_pw3(C.byref(a), stream()))
    return out, stats


def wgrad_pw3(dy, src, N, H, W, Cout, Cin, deferred=False):
    from starcop_amd._lib import sc_wgrad_pending
    lib = _lib.load()
    a = sc_wgrad_args()
    a.dy, a.nsrc = dy, 1
    a.src[0] = src
    a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, Cout, Cin, 1
    n = lib.sc_wgrad_pw3_workspace_floats(N, H, W, Cout, Cin)
    ws = torch.empty(n, device=DEV)
    a.part, a.part_floats = ws.data_ptr(), n
    dw = torch.full((Cout, Cin, 1, 1), float("nan"), device=DEV)
    a.dw = dw.data_ptr()
    _KEEP.append(ws)
    if not deferred:
        check(lib.sc_conv1x1_wgrad_pw3(C.byref(a), None, stream()))
        return dw
    pend = sc_wgrad_pending()
    check(lib.sc_conv1x1_wgrad_pw3(C.byref(a), C.byref(pend), stream()))
    import numpy as np
    descs = torch.from_numpy(np.frombuffer(bytes(pend), dtype=np.uint8).copy()).to(DEV)
    starts = torch.zeros(1, dtype=torch.int32, device=DEV)
    _KEEP.extend([descs, starts])
    check(lib.sc_wgrad_reduce_batch(ptr(descs), ptr(starts), 1, -(-int(pend.total) // 256), stream()))
    return dw


def pack_sp(w, cup, batched=False, terms=None):
    """decoder conv1 filter (Cout, Cup + Cskip, 3, 3) -> the phase / parity layout of sc_conv3x3_sp; batched: through the one-launch
    pack (the network's path)"""
    import numpy as np
    from starcop_amd._lib import PACK_SP
    lib = _lib.load()
    co, ci = w.shape[0], w.shape[1]
    from starcop_amd._lib import TERMS_F16X2
    terms = TERMS_F16X2 if terms is None else terms
    tfl = 4 if terms == 1 else 0                      # one bf16 term: half the entries
    out = torch.full((lib.sc_packed_weight_floats_sp(co, cup, ci - cup) // (2 if terms == 1 else 1),), float("nan"), device=DEV)     # every entry must be written
    if not batched:
        check(lib.sc_pack_weights_sp(ptr(w), ptr(out), co, cup, ci - cup, terms, stream()))
        return out
    total = lib.sc_pack_work_items(co, ci, 3, cup, tfl, PACK_SP)
    dt = np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("co_t", "<i4"),
                   ("tflip", "<i4"), ("bx3", "<i4"), ("total", "<u8")])
    descs = torch.from_numpy(np.array([(w.data_ptr(), out.data_ptr(), co, ci, 3, cup, tfl, PACK_SP, total)], dtype=dt).view(np.uint8).copy()).to(DEV)
    starts = torch.zeros(1, dtype=torch.int32, device=DEV)
    _KEEP.extend([descs, starts])
    check(lib.sc_pack_weights_batch(ptr(descs), ptr(starts), 1, -(-total // 256), stream()))
    return out


def conv_sp(srcs, wpk, N, H, W, Cout, want_stats=False, terms=None):
    """sc_conv3x3_sp: srcs = [half-resolution source (up = 1)] or [that, full-resolution skip source]; H x W = output size"""
    from starcop_amd._lib import TERMS_F16X2
    lib = _lib.load()
    a = sc_conv_args()
    a.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, Cout, 3, 32
    out = torch.full((N, Cout, H, W), float("nan"), device=DEV)
    a.out0, a.out1, a.csplit = out.data_ptr(), None, Cout
    a.accum0 = a.accum1 = 0
    a.add0 = a.add1 = None
    a.terms, a.down0, a.absmax = (TERMS_F16X2 if terms is None else terms), 0, None
    stats = torch.full((lib.sc_sp_stat_rows(N, H, W, Cout), Cout, 2), float("nan"), device=DEV) if want_stats else None
    a.stats = stats.data_ptr() if want_stats else None
    check(lib.sc_conv3x3_sp(C.byref(a), stream()))
    return out, stats


def pack_spd(w, cup, batched=False, vskip=False, terms=None, skip_tiles=False):
    """decoder conv1 filter (Cout, Cup + Cskip, 3, 3) -> the parity / tap layout of sc_conv3x3_sp_dgrad (vskip: with the skip
    channels as virtual channels of the tile's second half)"""
    import numpy as np
    from starcop_amd._lib import PACK_SPD
    lib = _lib.load()
    co, ci = w.shape[0], w.shape[1]
    from starcop_amd._lib import TERMS_F16X2
    terms = TERMS_F16X2 if terms is None else terms
    mode = 2 if skip_tiles else (1 if vskip else 0)          # sc_pack_weights_spd's vskip argument; the batch pack's transpose_flip = mode + 1
    tfl = (mode + 1) | (4 if terms == 1 else 0)
    out = torch.full((lib.sc_packed_weight_floats_spd(co, cup, ci - cup if skip_tiles else 0) // (2 if terms == 1 else 1),), float("nan"), device=DEV)
    if not batched:
        check(lib.sc_pack_weights_spd(ptr(w), ptr(out), co, ci, cup, mode, terms, stream()))
        return out
    total = lib.sc_pack_work_items(co, ci, 3, cup, tfl, PACK_SPD)
    dt = np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("co_t", "<i4"),
                   ("tflip", "<i4"), ("bx3", "<i4"), ("total", "<u8")])
    descs = torch.from_numpy(np.array([(w.data_ptr(), out.data_ptr(), co, ci, 3, cup, tfl, PACK_SPD, total)], dtype=dt).view(np.uint8).copy()).to(DEV)
    starts = torch.zeros(1, dtype=torch.int32, device=DEV)
    _KEEP.extend([descs, starts])
    check(lib.sc_pack_weights_batch(ptr(descs), ptr(starts), 1, -(-total // 256), stream()))
    return out


def conv_sp_dgrad(dy_src, wpk, N, H, W, Cup, absmax=None, accum_into=None, cskip=0, skip_into=None, terms=None):
    """sc_conv3x3_sp_dgrad: dy_src = the BNBWD operand of the layer's output (H x W) -> gradient of the half-resolution source;
    cskip > 0 (vskip pack): also the skip channels' full-resolution gradient -> returns (dprev, dskip)"""
    from starcop_amd._lib import TERMS_F16X2
    lib = _lib.load()
    a = sc_conv_args()
    a.nsrc = 1
    a.src[0] = dy_src
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, Cup + cskip, 3, 32
    out = accum_into if accum_into is not None else torch.full((N, Cup, H // 2, W // 2), float("nan"), device=DEV)
    osk = None
    if cskip:
        osk = skip_into if skip_into is not None else torch.full((N, cskip, H, W), float("nan"), device=DEV)
    a.out0, a.out1, a.csplit = out.data_ptr(), (osk.data_ptr() if cskip else None), Cup
    a.accum0, a.accum1 = (1 if accum_into is not None else 0), (1 if skip_into is not None else 0)
    a.add0 = a.add1 = None
    a.stats = None
    a.terms, a.down0 = (TERMS_F16X2 if terms is None else terms), 0
    a.absmax = absmax.data_ptr() if absmax is not None else None
    check(lib.sc_conv3x3_sp_dgrad(C.byref(a), stream()))
    return (out, osk) if cskip else out


def wgrad_sp(dy, src_lo, N, H, W, Cout, CinTotal, absmax=None, dw=None):
    """sc_conv3x3_sp_wgrad: the up-sampled channels' columns of the (Cout, CinTotal, 3, 3) gradient"""
    from starcop_amd._lib import TERMS_F16X2
    lib = _lib.load()
    a = sc_wgrad_args()
    a.dy = dy
    a.nsrc = 1
    a.src[0] = src_lo
    a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, Cout, CinTotal, 3
    a.terms = TERMS_F16X2
    a.absmax = absmax.data_ptr() if absmax is not None else None
    a.part, a.part_floats = None, 0
    if dw is None:
        dw = torch.full((Cout, CinTotal, 3, 3), float("nan"), device=DEV)
    a.dw = dw.data_ptr()
    nb = lib.sc_sp_wgrad_workspace_bytes(N, H, W, Cout, src_lo.C)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _KEEP.append(ws)
    check(lib.sc_conv3x3_sp_wgrad(C.byref(a), ptr(ws), nb, stream()))
    return dw
