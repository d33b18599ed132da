"""Per-kernel parity: every HIP op, called through the C ABI, against plain torch fp32 CPU ops on the
same seeded inputs (tolerance 1e-4 relative to the max magnitude unless stated; integer outputs bit-exact)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_ops import (dev, DEV, conv_mfma, conv_sp, conv_sp_dgrad, cst_affine, pack, pack_bx3, pack_sp, pack_spd, relerr, wgrad_mfma, wgrad_sp)  # noqa: E402
from starcop_amd import _lib  # noqa: E402
from starcop_amd._lib import (ACT_NONE, ACT_RELU, ACT_RELU6, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_NORM, SRC_RAW, STAT_BNBWD, STAT_CONV1,
                              STAT_DW, STAT_STEM, check, make_src, ptr, stream)  # noqa: E402

@pytest.fixture(params=[3, 4], ids=["bf16x3", "fp16x2"])
def split_mode(request):
    """run a split-kernel test under both operand splits (three bf16 terms / two fp16 terms)"""
    import hip_ops
    old, hip_ops.DEFAULT_BX3_TERMS = hip_ops.DEFAULT_BX3_TERMS, request.param
    yield request.param
    hip_ops.DEFAULT_BX3_TERMS = old


TOL = 1e-4


def act_ref(v, act):
    return v if act == ACT_NONE else (F.relu(v) if act == ACT_RELU else F.relu6(v))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("ks,cin,cout,co_t,H,W", [
    (3, 16, 16, 32, 32, 32), (3, 16, 16, 16, 32, 32), (3, 40, 16, 16, 12, 40), (3, 32, 8, 16, 36, 70), (3, 32, 64, 64, 32, 64), (3, 24, 96, 32, 12, 40), (3, 64, 40, 64, 36, 32),
    (1, 16, 96, 32, 16, 16), (1, 96, 24, 32, 24, 20), (1, 160, 128, 64, 16, 16), (1, 24, 144, 32, 10, 13),
    (1, 320, 1280, 64, 2, 3), (1, 960, 160, 64, 2, 3), (3, 1376, 256, 64, 4, 6)])
def test_conv_mfma_fwd_affine_stats(hip, ks, cin, cout, co_t, H, W):
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, ks, ks, seed=2, scale=0.2)
    sc, sh = rnd(cin, seed=3) * 0.5 + 1.0, rnd(cin, seed=4) * 0.3
    act = ACT_RELU6
    ref = F.conv2d(act_ref(x * sc[None, :, None, None] + sh[None, :, None, None], act), w, padding=ks // 2)
    xd, wd = dev(x), dev(w)
    src = make_src(xd, cin, SRC_AFFINE, act=act, cst=cst_affine(sc, sh))
    (out,), stats = conv_mfma([src], pack(wd, co_t, 0), N, H, W, cout, ks, co_t, want_stats=True)
    assert relerr(out, ref) < TOL
    st = stats.double().sum(0).cpu()
    assert relerr(st[:, 0], ref.double().sum((0, 2, 3))) < 1e-4
    assert relerr(st[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-4


def test_conv_mfma_upsample_concat(hip):
    """decoder conv1: input = cat(nearest_up2(prev), skip) (smp DecoderBlock.forward)."""
    N, c0, c1, cout, H, W = 2, 24, 8, 48, 16, 64
    prev, skip = rnd(N, c0, H // 2, W // 2, seed=1), rnd(N, c1, H, W, seed=2)
    w = rnd(cout, c0 + c1, 3, 3, seed=3, scale=0.1)
    sc0, sh0 = rnd(c0, seed=4) * 0.3 + 1, rnd(c0, seed=5) * 0.2
    xin = torch.cat([F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]),
                                   scale_factor=2, mode="nearest"), skip], 1)
    ref = F.conv2d(xin, w, padding=1)
    s0 = make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))
    s1 = make_src(dev(skip), c1, SRC_RAW)
    (out,), _ = conv_mfma([s0, s1], pack(dev(w), 32, 0), N, H, W, cout, 3, 32)
    assert relerr(out, ref) < TOL


@pytest.mark.parametrize("ks,cin,cout,H,W", [(3, 32, 16, 32, 32), (3, 16, 16, 32, 64), (3, 16, 32, 20, 40), (3, 80, 32, 16, 32), (1, 96, 16, 16, 16), (1, 24, 144, 8, 16),
                                              (1, 320, 1280, 2, 3), (3, 256, 128, 4, 6)])
def test_conv_dgrad_bnbwd_split_add(hip, ks, cin, cout, H, W):
    """backward-data = same kernel on the transposed+flipped filter; dy formed on load from (g, y)."""
    N = 2
    g, y = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    w = rnd(cout, cin, ks, ks, seed=3, scale=0.2)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6), rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where(yh > 0, g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    ref = F.conv_transpose2d(dy, w, padding=ks // 2)
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    src = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(y))
    co_t = 16 if (cin <= 16 and ks == 3) else (32 if cin <= 32 else 64)
    wpk = pack(dev(w), co_t, 1)
    (out,), _ = conv_mfma([src], wpk, N, H, W, cin, ks, co_t)
    assert relerr(out, ref) < TOL
    # epilogue adds + accumulate
    add0, old = rnd(N, cin, H, W, seed=9), rnd(N, cin, H, W, seed=10)
    o = dev(old).clone()
    conv_mfma([src], wpk, N, H, W, cin, ks, co_t, add0=dev(add0), accum=(1, 0), outs=[o])
    assert relerr(o, ref + add0 + old) < TOL
    if co_t == 16:
        return          # the thin-layer kernel has a single output
    # channel split
    cs = cin // 2 if (cin // 2) % 8 == 0 else 8
    outs, _ = conv_mfma([src], wpk, N, H, W, cin, ks, co_t, csplit=cs)
    assert relerr(outs[0], ref[:, :cs]) < TOL and relerr(outs[1], ref[:, cs:]) < TOL


@pytest.mark.parametrize("cin,cout,co_t,H,W", [(384, 64, 64, 8, 8), (960, 160, 64, 4, 4), (576, 96, 32, 8, 6), (24, 144, 32, 10, 13), (1280, 320, 64, 2, 3), (40, 8, 32, 5, 7)])
def test_conv1x1_ksplit(hip, cin, cout, co_t, H, W):
    """low-resolution pointwise layers: 32-pixel tiles, K split over the four waves; forward (affine + stats) and dgrad epilogues"""
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 1, 1, seed=2, scale=0.2)
    sc, sh = rnd(cin, seed=3) * 0.5 + 1.0, rnd(cin, seed=4) * 0.3
    ref = F.conv2d(F.relu6(x * sc[None, :, None, None] + sh[None, :, None, None]), w)
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU6, cst=cst_affine(sc, sh))
    wpk = pack(dev(w), co_t, 0)
    (out,), stats = conv_mfma([src], wpk, N, H, W, cout, 1, co_t, want_stats=True, ksplit=True)
    assert relerr(out, ref) < TOL
    st = stats.double().sum(0).cpu()
    assert relerr(st[:, 0], ref.double().sum((0, 2, 3))) < 1e-4
    assert relerr(st[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-4
    # raw source
    (out,), _ = conv_mfma([make_src(dev(x), cin, SRC_RAW)], wpk, N, H, W, cout, 1, co_t, ksplit=True)
    assert relerr(out, F.conv2d(x, w)) < TOL
    # backward-data form: BNBWD source, transposed filter, residual add + accumulate, channel split
    g, y = rnd(N, cout, H, W, seed=5), rnd(N, cout, H, W, seed=6)
    a, b = rnd(cout, seed=7) * 0.2 + 1, rnd(cout, seed=8) * 0.2
    A, B, D = rnd(cout, seed=9), rnd(cout, seed=10) * 0.1, rnd(cout, seed=11) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where((yh > 0) & (yh < 6), g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    refb = F.conv_transpose2d(dy, w)
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    srcb = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cst), aux=dev(y))
    cb = 32 if cin <= 32 else 64
    wb = pack(dev(w), cb, 1)
    add0, old = rnd(N, cin, H, W, seed=12), rnd(N, cin, H, W, seed=13)
    o = dev(old).clone()
    conv_mfma([srcb], wb, N, H, W, cin, 1, cb, add0=dev(add0), accum=(1, 0), outs=[o], ksplit=True)
    assert relerr(o, refb + add0 + old) < TOL
    cs = 8
    outs, _ = conv_mfma([srcb], wb, N, H, W, cin, 1, cb, csplit=cs, ksplit=True)
    assert relerr(outs[0], refb[:, :cs]) < TOL and relerr(outs[1], refb[:, cs:]) < TOL


def test_pack_weights_batch(hip):
    """one launch packs every layer: identical bytes to the per-layer entry points (fp32 and split-bf16 layouts, fwd and dgrad)"""
    import numpy as np
    lib = hip
    cases = [(96, 16, 1, 32, 0, 0), (96, 16, 1, 32, 1, 0), (24, 40, 3, 32, 0, 0), (16, 32, 3, 16, 0, 0), (64, 152, 3, 64, 0, 3), (64, 152, 3, 64, 1, 3), (32, 80, 3, 32, 1, 3), (64, 152, 3, 64, 0, 1)]
    dt = np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("co_t", "<i4"), ("tflip", "<i4"), ("bx3", "<i4"), ("total", "<u8")])
    rows, starts, nblk, want, got = [], [], 0, [], []
    for k, (co, ci, ks, cot, tf, bx) in enumerate(cases):
        w = dev(rnd(co, ci, ks, ks, seed=k))
        ref = pack_bx3(w, cot, tf, bx) if bx else pack(w, cot, tf)
        out = torch.full_like(ref, float("nan")); _ = dev  # noqa
        want.append(ref); got.append(out)
        total = lib.sc_pack_work_items(co, ci, ks, cot, tf, bx)
        rows.append((w.data_ptr(), out.data_ptr(), co, ci, ks, cot, tf, bx, total))       # bx = number of bf16 terms (0: fp32 layout)
        starts.append(nblk); nblk += -(-total // 256)
    descs = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).copy()).to(DEV)
    st_t = torch.tensor(starts, dtype=torch.int32).to(DEV)
    check(lib.sc_pack_weights_batch(ptr(descs), ptr(st_t), len(rows), nblk, stream()))
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


# ---- split-bf16 (three-term) 3x3 convolution: fp32 accuracy on the bf16 matrix cores -------------------------------
BX3_TOL = 1e-5      # vs an fp64 reference; each case is also required to be no worse than 3x the fp32 MFMA kernel's error


@pytest.mark.parametrize("cin,cout,co_t,H,W", [(16, 64, 64, 32, 32), (40, 16, 32, 12, 40), (32, 64, 64, 36, 70), (24, 96, 32, 20, 40),
                                                (64, 40, 64, 37, 33), (1376, 256, 64, 4, 6), (8, 8, 32, 8, 8)])
def test_conv_bx3_fwd_affine_stats(hip, split_mode, cin, cout, co_t, H, W):
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    sc, sh = rnd(cin, seed=3) * 0.5 + 1.0, rnd(cin, seed=4) * 0.3
    act = ACT_RELU6
    xin = act_ref(x * sc[None, :, None, None] + sh[None, :, None, None], act)      # fp32, as the kernel's prologue
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    src = make_src(dev(x), cin, SRC_AFFINE, act=act, cst=cst_affine(sc, sh))
    (out,), stats = conv_mfma([src], pack_bx3(dev(w), co_t, 0), N, H, W, cout, 3, co_t, want_stats=True, bx3=True)
    assert relerr(out, ref) < BX3_TOL
    # not worse than the fp32 MFMA kernel on the same inputs
    co32 = 32 if co_t == 32 else 64
    (out32,), _ = conv_mfma([src], pack(dev(w), co32, 0), N, H, W, cout, 3, co32)
    assert relerr(out, ref) < 3 * relerr(out32, ref) + 1e-7
    st = stats.double().sum(0).cpu()
    assert relerr(st[:, 0], ref.sum((0, 2, 3))) < 1e-5
    assert relerr(st[:, 1], (ref ** 2).sum((0, 2, 3))) < 1e-5


def test_conv_bx3_upsample_concat(hip, split_mode):
    N, c0, c1, cout, H, W = 2, 32, 24, 48, 16, 64
    prev, skip = rnd(N, c0, H // 2, W // 2, seed=1), rnd(N, c1, H, W, seed=2)
    w = rnd(cout, c0 + c1, 3, 3, seed=3, scale=0.1)
    sc0, sh0 = rnd(c0, seed=4) * 0.3 + 1, rnd(c0, seed=5) * 0.2
    xin = torch.cat([F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]),
                                   scale_factor=2, mode="nearest"), skip], 1)
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    s0 = make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))
    s1 = make_src(dev(skip), c1, SRC_RAW)
    for co_t in (32, 64):
        (out,), _ = conv_mfma([s0, s1], pack_bx3(dev(w), co_t, 0), N, H, W, cout, 3, co_t, bx3=True)
        assert relerr(out, ref) < BX3_TOL


# ---- decoder conv1 as a sub-pixel convolution (conv_sp.hip): conv3x3(cat([nearest_up2(prev), skip])) from the low-resolution prev ----
@pytest.mark.parametrize("c0,c1,cout,H,W", [(32, 24, 48, 16, 64), (64, 16, 32, 64, 128), (16, 0, 32, 32, 64), (48, 40, 64, 12, 20),
                                             (20, 9, 40, 36, 72), (256, 32, 128, 8, 16), (32, 0, 16, 96, 80), (1280, 96, 64, 4, 6)])
def test_conv_sp_matches_upsample_concat_conv(hip, c0, c1, cout, H, W):
    """the reference's op sequence (smp DecoderBlock: interpolate(nearest, x2) -> cat -> conv3x3) in float64 against sc_conv3x3_sp:
    low-resolution planes of both tile shapes (>= 32 wide: 8 x 32 tiles; narrower: 16 x 16), ragged tile edges, channel counts that
    are not multiples of the 16-channel chunk, with and without skip channels, BatchNorm + ReLU / raw prologues, statistics rows"""
    from starcop_amd._lib import TERMS_F16X2
    N = 2
    prev = rnd(N, c0, H // 2, W // 2, seed=1)
    w = rnd(cout, c0 + c1, 3, 3, seed=3, scale=0.1)
    sc0, sh0 = rnd(c0, seed=4) * 0.3 + 1, rnd(c0, seed=5) * 0.2
    up = F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2, mode="nearest")
    srcs = [make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))]
    if c1:
        skip = rnd(N, c1, H, W, seed=2)
        sc1, sh1 = rnd(c1, seed=6) * 0.3 + 1, rnd(c1, seed=7) * 0.2
        xin = torch.cat([up, skip * sc1[None, :, None, None] + sh1[None, :, None, None]], 1)
        srcs.append(make_src(dev(skip), c1, SRC_AFFINE, act=ACT_NONE, cst=cst_affine(sc1, sh1)))
    else:
        xin = up
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    wd = dev(w)
    for batched in (False, True):
        wpk = pack_sp(wd, c0, batched=batched)
        assert bool(torch.isfinite(wpk.view(torch.int16).float()).all())
        out, stats = conv_sp(srcs, wpk, N, H, W, cout, want_stats=True)
        assert relerr(out, ref) < BX3_TOL
        st = stats.double().sum(0).cpu()
        assert relerr(st[:, 0], ref.sum((0, 2, 3))) < 1e-5
        assert relerr(st[:, 1], (ref ** 2).sum((0, 2, 3))) < 1e-5
    # as close to float64 as the 3x3 form of the same arithmetic (two fp16 terms per operand)
    if c0 % 16 == 0 or not c1:           # (the 3x3 kernel's concat needs 16-channel-aligned first sources)
        co_t = 64 if cout > 32 else 32
        (out3,), _ = conv_mfma(srcs, pack_bx3(wd, co_t, 0, TERMS_F16X2), N, H, W, cout, 3, co_t, bx3=True, terms=TERMS_F16X2)
        assert relerr(out, ref) < 3 * relerr(out3, ref) + 1e-7


def test_conv_sp_raw_sources_and_filter_range(hip):
    """RAW sources (the identity constants table), and phase filters at the edge of the fp16 range: |w| up to 250 -> sums of four
    taps up to 1000, scaled by 2^6 (not the 3x3 kernels' 2^8) before the fp16 split"""
    N, c0, c1, cout, H, W = 1, 16, 16, 32, 16, 64
    prev, skip = rnd(N, c0, H // 2, W // 2, seed=11), rnd(N, c1, H, W, seed=12)
    w = rnd(cout, c0 + c1, 3, 3, seed=13, scale=0.1)
    w[0, 0] = 250.0
    w[1, 1] = -250.0
    ref = F.conv2d(torch.cat([F.interpolate(prev, scale_factor=2, mode="nearest"), skip], 1).double(), w.double(), padding=1)
    out, _ = conv_sp([make_src(dev(prev), c0, SRC_RAW, up=1), make_src(dev(skip), c1, SRC_RAW)], pack_sp(dev(w), c0), N, H, W, cout)
    assert relerr(out, ref) < BX3_TOL


def _sp_phase_reference(prev, skip, w, rnd_op):
    """conv3x3(cat([nearest_up2(prev), skip])) written as the sub-pixel convolution of conv_sp_pack.h, in float64, with `rnd_op`
    applied to the operands the kernel rounds: the activations and the PHASE filters (sums of taps, formed in fp32 in kh, kw order)"""
    N, cu, Hl, Wl = prev.shape
    S = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    out = torch.zeros(N, w.shape[0], 2 * Hl, 2 * Wl, dtype=torch.float64)
    xp = F.pad(rnd_op(prev), (1, 1, 1, 1))
    for py in range(2):
        for px in range(2):
            wph = torch.zeros(w.shape[0], cu, 2, 2)
            for a in range(2):
                for b in range(2):
                    for kh in S[(py, a)]:
                        for kw in S[(px, b)]:
                            wph[:, :, a, b] += w[:, :cu, kh, kw]
            # source offset of tap (a, b): (a - 1 + py, b - 1 + px) -> a 2x2 correlation on the padded plane shifted by (py, px)
            out[:, :, py::2, px::2] += F.conv2d(xp[:, :, py:py + Hl + 1, px:px + Wl + 1], rnd_op(wph))
    if skip is not None:
        out += F.conv2d(rnd_op(skip), rnd_op(w[:, cu:]), padding=1)
    return out


@pytest.mark.parametrize("c0,c1,cout,H,W", [(64, 16, 32, 16, 64), (128, 24, 64, 24, 80), (40, 9, 24, 36, 70), (32, 0, 33, 20, 72)])
def test_conv_sp_bf16_single_term(hip, c0, c1, cout, H, W):
    """terms = 1 of the sub-pixel kernels (the "bf16" precision mode): forward and data gradient equal float64 evaluations of the same
    sub-pixel sums on bf16-rounded operands (activations, dy and the PHASE filters) to fp32-accumulation accuracy, and stay within
    bf16 rounding of the unrounded layer; both pack paths, statistics rows, the one-launch skip gradient (vskip)"""
    N = 2
    prev, w = rnd(N, c0, H // 2, W // 2, seed=1), rnd(cout, c0 + c1, 3, 3, seed=3, scale=0.1)
    skip = rnd(N, c1, H, W, seed=2) if c1 else None
    rb = lambda t: t.float().bfloat16().double()
    ident = lambda t: t.double()
    full = F.conv2d(torch.cat([F.interpolate(prev, scale_factor=2, mode="nearest")] + ([skip] if c1 else []), 1).double(), w.double(), padding=1)
    assert relerr(_sp_phase_reference(prev, skip, w, ident), full) < 1e-6          # (the restated sums are the layer)
    ref_r = _sp_phase_reference(prev, skip, w, rb)
    srcs = [make_src(dev(prev), c0, SRC_RAW, up=1)] + ([make_src(dev(skip), c1, SRC_RAW)] if c1 else [])
    wd = dev(w)
    for batched in (False, True):
        wpk = pack_sp(wd, c0, batched=batched, terms=1)
        assert bool(torch.isfinite(wpk.view(torch.int16).float()).all())
        out, stats = conv_sp(srcs, wpk, N, H, W, cout, want_stats=True, terms=1)
        assert relerr(out, ref_r) < 5e-6 and relerr(out, full) < 2e-2
        assert relerr(stats.double().sum(0).cpu()[:, 1], (ref_r ** 2).sum((0, 2, 3))) < 1e-5
    # data gradient: dy = g exactly (identity BatchNorm-backward constants, no activation), so the rounded operand is rb(g)
    if c0 % 32:
        return
    g = rnd(N, cout, H, W, seed=5)
    cst = torch.zeros(cout, SC_CST)
    cst[:, 0], cst[:, 2] = 1.0, 1.0
    src = make_src(dev(g), cout, SRC_BNBWD, act=ACT_NONE, cst=dev(cst), aux=dev(rnd(N, cout, H, W, seed=6)))
    pv = prev.double().requires_grad_(True)
    sk = skip.double().requires_grad_(True) if c1 else None
    _sp_phase_reference(pv, sk, w, lambda t: t.double() if t.requires_grad or t.grad_fn is not None else rb(t)).backward(rb(g))
    for batched in (False, True):
        dx = conv_sp_dgrad(src, pack_spd(wd, c0, batched=batched, terms=1), N, H, W, c0, terms=1)
        assert relerr(dx, pv.grad) < 5e-6
    if c0 <= 64 and 0 < c1 <= 16:
        for batched in (False, True):
            o_up, o_sk = conv_sp_dgrad(src, pack_spd(wd, c0, batched=batched, vskip=True, terms=1), N, H, W, c0, cskip=c1, terms=1)
            assert relerr(o_up, pv.grad) < 5e-6 and relerr(o_sk, sk.grad) < 5e-6


@pytest.mark.parametrize("cin,cout,H,W", [(32, 16, 32, 32), (16, 32, 20, 40), (80, 32, 16, 32), (256, 128, 4, 6), (152, 64, 24, 32)])
def test_conv_bx3_dgrad_bnbwd_split_add(hip, split_mode, cin, cout, H, W):
    N = 2
    g, y = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    w = rnd(cout, cin, 3, 3, seed=3, scale=0.2)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6), rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where(yh > 0, g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    src = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(y))
    co_t = 32 if cin <= 32 else 64
    wpk = pack_bx3(dev(w), co_t, 1)
    (out,), _ = conv_mfma([src], wpk, N, H, W, cin, 3, co_t, bx3=True)
    assert relerr(out, ref) < 1e-5          # dy itself is formed in fp32 (fma contraction differs from the host's)
    add0, old = rnd(N, cin, H, W, seed=9), rnd(N, cin, H, W, seed=10)
    o = dev(old).clone()
    conv_mfma([src], wpk, N, H, W, cin, 3, co_t, add0=dev(add0), accum=(1, 0), outs=[o], bx3=True)
    assert relerr(o, ref + add0.double() + old.double()) < 1e-5
    cs = cin // 2 if (cin // 2) % 8 == 0 else 8
    outs, _ = conv_mfma([src], wpk, N, H, W, cin, 3, co_t, csplit=cs, bx3=True)
    assert relerr(outs[0], ref[:, :cs]) < 1e-5 and relerr(outs[1], ref[:, cs:]) < 1e-5


@pytest.mark.parametrize("ks,cin,cout,H,W,two", [(3, 16, 16, 32, 32, False), (3, 32, 16, 36, 70, True), (3, 48, 8, 20, 32, False), (3, 32, 64, 16, 32, True), (3, 80, 48, 20, 36, True),
                                                  (3, 64, 16, 8, 64, False), (1, 96, 24, 16, 16, False), (1, 24, 144, 16, 24, False),
                                                  (1, 160, 320, 8, 8, False), (1, 320, 1280, 2, 3, False), (3, 256, 256, 4, 6, False)])
def test_conv_wgrad(hip, ks, cin, cout, H, W, two):
    N = 3
    g, y = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6), rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where((yh > 0) & (yh < 6), g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    dys = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cst), aux=dev(y))
    if two:
        c0 = cin - 8
        prev, skip = rnd(N, c0, H // 2, W // 2, seed=11), rnd(N, 8, H, W, seed=12)
        sc0, sh0 = rnd(c0, seed=13) * 0.3 + 1, rnd(c0, seed=14) * 0.2
        xin = torch.cat([F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2), skip], 1)
        srcs = [make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0)), make_src(dev(skip), 8, SRC_RAW)]
    else:
        xin = rnd(N, cin, H, W, seed=11)
        srcs = [make_src(dev(xin), cin, SRC_RAW)]
    xin = xin.clone().requires_grad_(False)
    w = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    F.conv2d(xin, w, padding=ks // 2).backward(dy)
    dw = wgrad_mfma(dys, srcs, N, H, W, cout, cin, ks)
    assert relerr(dw, w.grad) < TOL


@pytest.mark.parametrize("cin,cout,co_t,H,W", [(32, 64, 64, 36, 70), (24, 96, 32, 20, 40), (288, 128, 64, 8, 12)])
def test_conv_bf16_single_term(hip, cin, cout, co_t, H, W):
    """terms = 1 ("bf16" precision mode): equals an fp64 convolution of the bf16-rounded operands to fp32-accumulation accuracy,
    and is within bf16 rounding of the unrounded result; forward, dgrad and wgrad."""
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    rb = lambda t: t.bfloat16().double()
    ref_r = F.conv2d(rb(x), rb(w), padding=1)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    src = make_src(dev(x), cin, SRC_RAW)
    (out,), _ = conv_mfma([src], pack_bx3(dev(w), co_t, 0, 1), N, H, W, cout, 3, co_t, bx3=True, terms=1)
    assert relerr(out, ref_r) < 5e-6 and relerr(out, ref) < 2e-2
    g = rnd(N, cout, H, W, seed=3)
    cb = 32 if cin <= 32 else 64
    (dx,), _ = conv_mfma([make_src(dev(g), cout, SRC_RAW)], pack_bx3(dev(w), cb, 1, 1), N, H, W, cin, 3, cb, bx3=True, terms=1)
    assert relerr(dx, F.conv_transpose2d(rb(g), rb(w), padding=1)) < 5e-6
    if cin > 32:
        wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(rb(x), wz, padding=1).backward(rb(g))
        dw = wgrad_mfma(make_src(dev(g), cout, SRC_RAW), [src], N, H, W, cout, cin, 3, bx3=True, terms=1)
        assert relerr(dw, wz.grad) < 5e-6


@pytest.mark.parametrize("cin,cout,co_t,H,W", [(32, 64, 64, 36, 70), (80, 32, 32, 20, 40), (288, 128, 64, 8, 12)])
def test_conv_two_term_split(hip, cin, cout, co_t, H, W):
    """terms = 2 (the "fp32-bwd2" / "fp32-2" modes): a0*b0 + a0*b1 + a1*b0 of the exact two-term bf16 splits -- equals an
    fp64 convolution of the 16-bit-significand operands (minus the a1*b1 term, 2^-18) and is within 2e-5 of the unrounded
    fp64 result (operand error 2^-18, random sign, averaged over K >= 288 terms); forward, dgrad and wgrad with the
    network's prologues."""
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    cst = torch.rand(cin, SC_CST, generator=torch.Generator().manual_seed(12)) + 0.5
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, cst=dev(cst))
    xin = torch.relu(x.double() * cst[:, 0].double()[None, :, None, None] + cst[:, 1].double()[None, :, None, None])
    ref = F.conv2d(xin, w.double(), padding=1)
    (out,), _ = conv_mfma([src], pack_bx3(dev(w), co_t, 0, 2), N, H, W, cout, 3, co_t, bx3=True, terms=2)
    (out3,), _ = conv_mfma([src], pack_bx3(dev(w), co_t, 0, 3), N, H, W, cout, 3, co_t, bx3=True, terms=3)
    e2, e3 = relerr(out, ref), relerr(out3, ref)
    assert e2 < 2e-5 and e3 < 4e-6, (e2, e3)        # e3: the fp32 accumulation noise of the K = 2592 reduction
    g = rnd(N, cout, H, W, seed=3)
    cb = 32 if cin <= 32 else 64
    (dx,), _ = conv_mfma([make_src(dev(g), cout, SRC_RAW)], pack_bx3(dev(w), cb, 1, 2), N, H, W, cin, 3, cb, bx3=True, terms=2)
    assert relerr(dx, F.conv_transpose2d(g.double(), w.double(), padding=1)) < 2e-5
    if cin >= 32 and cout >= 32:
        wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, wz, padding=1).backward(g.double())
        dw = wgrad_mfma(make_src(dev(g), cout, SRC_RAW), [src], N, H, W, cout, cin, 3, bx3=True, terms=2)
        assert relerr(dw, wz.grad) < 2e-5


@pytest.mark.parametrize("gscale", [1.0, 3e-8, 2e4])
@pytest.mark.parametrize("cin,cout,co_t,H,W", [(32, 64, 64, 36, 70), (80, 32, 32, 20, 40), (288, 128, 64, 8, 12)])
def test_conv_two_fp16_terms(hip, cin, cout, co_t, H, W, gscale):
    """terms = SC_TERMS_F16X2: two fp16 terms per operand (22 significand bits), three products, exact power-of-two range
    scaling -- as accurate as the three-term bf16 split (<= 2e-6 of fp64) whatever the magnitude of the gradient tensor
    (the scale comes from the max |A g| that sc_bn_bwd_reduce leaves in `absmax`); forward, dgrad and wgrad."""
    from starcop_amd._lib import TERMS_F16X2
    N = 2
    x, w = rnd(N, cin, H, W, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    w[0, 0, 0, 0], w[1, 1, 1, 1] = 3e-4, 40.0                      # tiny and huge filter entries
    cst = torch.rand(cin, SC_CST, generator=torch.Generator().manual_seed(11)) + 0.5     # (seeded: the unseeded draw made the bound below flaky)
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, cst=dev(cst))
    xin = torch.relu(x.double() * cst[:, 0].double()[None, :, None, None] + cst[:, 1].double()[None, :, None, None])
    ref = F.conv2d(xin, w.double(), padding=1)
    (out,), st = conv_mfma([src], pack_bx3(dev(w), co_t, 0, TERMS_F16X2), N, H, W, cout, 3, co_t, bx3=True, terms=TERMS_F16X2, want_stats=True)
    # 9 * cin fp32 accumulations: 2e-6 of the largest output up to 80 input channels, 3e-6 at 288 (unseeded draws reached 2.7e-6)
    assert relerr(out, ref) < (2e-6 if cin <= 80 else 3e-6)
    assert relerr(st.sum(0)[:, 0], ref.sum((0, 2, 3))) < 1e-5       # the statistics see the un-scaled values
    # gradient operand through the BatchNorm-backward prologue, at three magnitudes
    g, y = rnd(N, cout, H, W, seed=3) * gscale, rnd(N, cout, H, W, seed=4)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = a.clone(), rnd(cout, seed=7) * 0.1 * gscale, rnd(cout, seed=8) * 0.1 * gscale
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    gm = torch.where(yh > 0, g, torch.zeros(()))
    dy = (gm.double() * A.double()[None, :, None, None] + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None])
    cstb = torch.zeros(cout, SC_CST); cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = a, b, A, B, D
    # the range hint exactly as the network produces it: sc_bn_bwd_reduce on (g, y) with the forward constants (scale = A)
    cf = torch.zeros(cout, SC_CST); cf[:, 0], cf[:, 1], cf[:, 2], cf[:, 3] = a, b, 0.0, 1.0
    rows = hip.sc_stat_rows(STAT_BNBWD, N, H, W)
    sums = torch.empty(rows * cout * 2, dtype=torch.float64, device=DEV)
    amax = torch.zeros(1, device=DEV)
    aact = torch.full((1,), 0.25, device=DEV)            # the activation record is sticky: it is only ever raised
    gd, yd = dev(g), dev(y)
    check(hip.sc_bn_bwd_reduce(ptr(gd), ptr(yd), ptr(dev(cf)), ACT_RELU, ptr(sums), N, cout, H * W, ptr(amax), ptr(aact), stream()))
    want_max = float((gm.abs().amax((0, 2, 3)) * a.abs()).max())
    assert float(amax) == pytest.approx(want_max, rel=1e-6)
    assert float(aact) == pytest.approx(float(yh.abs().max()), rel=1e-6)     # max |BatchNorm output| >= max |relu(.)|
    dsrc = make_src(gd, cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cstb), aux=yd)
    cb = 32 if cin <= 32 else 64
    (dx,), _ = conv_mfma([dsrc], pack_bx3(dev(w), cb, 1, TERMS_F16X2), N, H, W, cin, 3, cb, bx3=True, terms=TERMS_F16X2, absmax=amax)
    assert relerr(dx, F.conv_transpose2d(dy, w.double(), padding=1)) < 1e-5     # dy itself is formed in fp32
    if cin >= 32 and cout >= 32:
        wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, wz, padding=1).backward(dy)
        dw = wgrad_mfma(dsrc, [src], N, H, W, cout, cin, 3, bx3=True, terms=TERMS_F16X2, absmax=amax)
        dw3 = wgrad_mfma(dsrc, [src], N, H, W, cout, cin, 3, bx3=True, terms=3)
        assert relerr(dw, wz.grad) < max(1e-5, 2 * relerr(dw3, wz.grad))
    # bn_bwd_small leaves the same hint
    amax2, aact2 = torch.zeros(1, device=DEV), torch.full((1,), 1e9, device=DEV)
    dg2, db2, cb2 = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), torch.empty(cout, SC_CST, device=DEV)
    check(hip.sc_bn_bwd_small(ptr(gd), ptr(yd), ptr(dev(cf)), ACT_RELU, N, cout, H * W, ptr(dg2), ptr(db2), ptr(cb2), ptr(amax2), ptr(aact2), stream()))
    assert float(amax2) == float(amax) and float(aact2) == 1e9              # a larger record is never lowered


@pytest.mark.parametrize("cin,cout,cs,H,W,co_t", [(80, 32, 64, 16, 64, 64), (152, 64, 128, 24, 40, 64), (32, 16, 32, 20, 36, 32), (288, 128, 256, 8, 12, 64)])
def test_conv_bx3_dgrad_fused_upsample_backward(hip, split_mode, cin, cout, cs, H, W, co_t):
    """decoder conv1 data gradient: channels [0, cs) belong to the nearest-x2-upsampled input -> stored as 2x2 sums at half
    resolution (== F.interpolate backward), the skip channels at full resolution; with and without accumulation."""
    N = 2
    g = rnd(N, cout, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=3, scale=0.2)
    ref = F.conv_transpose2d(g.double(), w.double(), padding=1)
    ref_up = F.avg_pool2d(ref[:, :cs], 2) * 4
    src = make_src(dev(g), cout, SRC_RAW)
    wpk = pack_bx3(dev(w), co_t, 1)
    if cs < cin:
        outs, _ = conv_mfma([src], wpk, N, H, W, cin, 3, co_t, csplit=cs, bx3=True, down0=True)
        assert relerr(outs[0], ref_up) < 1e-5 and relerr(outs[1], ref[:, cs:]) < 1e-5
    old = rnd(N, cs, H // 2, W // 2, seed=5)
    o0 = dev(old).clone()
    extra = [torch.empty(N, cin - cs, H, W, device=DEV)] if cs < cin else []
    conv_mfma([src], wpk, N, H, W, cin, 3, co_t, csplit=cs, bx3=True, down0=True, accum=(1, 0), outs=[o0] + extra)
    assert relerr(o0, ref_up + old.double()) < 1e-5


@pytest.mark.parametrize("cup,csk,cout,H,W", [(64, 16, 32, 16, 64), (128, 24, 64, 24, 80), (32, 0, 16, 20, 72), (256, 32, 128, 8, 12),
                                               (40, 9, 24, 36, 70), (200, 8, 48, 64, 64), (1280, 96, 40, 4, 6)])
def test_conv_sp_dgrad_matches_upsample_backward(hip, cup, csk, cout, H, W):
    """gradient of conv3x3(cat([nearest_up2(prev), skip])) w.r.t. prev -- autograd of the reference's op sequence in float64 (conv
    transpose, then F.interpolate's backward = 2x2 sums) -- against sc_conv3x3_sp_dgrad: a stride-2 4x4 convolution of dy at half
    resolution; BatchNorm / ReLU-backward operand with a range hint, plain operand, accumulation, both pack paths, both tile shapes"""
    N = 2
    g, yraw = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    w = rnd(cout, cup + csk, 3, 3, seed=3, scale=0.2)
    cst = torch.zeros(cout, SC_CST)
    cst[:, 0], cst[:, 1] = rnd(cout, seed=4) * 0.3 + 1, rnd(cout, seed=5) * 0.2          # forward BatchNorm scale / shift
    cst[:, 2], cst[:, 3], cst[:, 4] = rnd(cout, seed=6) * 0.5 + 1, rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.05   # A, B, D
    yh = yraw * cst[:, 0][None, :, None, None] + cst[:, 1][None, :, None, None]
    dy = torch.where(yh > 0, g, torch.zeros_like(g)) * cst[:, 2][None, :, None, None] + yraw * cst[:, 3][None, :, None, None] + cst[:, 4][None, :, None, None]
    ref = F.avg_pool2d(F.conv_transpose2d(dy.double(), w.double(), padding=1)[:, :cup], 2) * 4
    amax = torch.tensor([float((cst[:, 2][None, :, None, None] * g).abs().max())], device=DEV)
    src = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(yraw))
    wd = dev(w)
    for batched in (False, True):
        wpk = pack_spd(wd, cup, batched=batched)
        assert bool(torch.isfinite(wpk.view(torch.int16).float()).all())
        out = conv_sp_dgrad(src, wpk, N, H, W, cup, absmax=amax)
        assert relerr(out, ref) < BX3_TOL
    old = rnd(N, cup, H // 2, W // 2, seed=9)
    o0 = dev(old).clone()
    conv_sp_dgrad(src, wpk, N, H, W, cup, absmax=amax, accum_into=o0)
    assert relerr(o0, ref + old.double()) < BX3_TOL
    # <= 64 up-sampled and <= 16 skip channels (smp's decoder.blocks.3): the skip channels' full-resolution gradient from the same launch
    if cup <= 64 and 0 < csk <= 16:
        ref_sk = F.conv_transpose2d(dy.double(), w.double(), padding=1)[:, cup:]
        for batched in (False, True):
            wv = pack_spd(wd, cup, batched=batched, vskip=True)
            o_up, o_sk = conv_sp_dgrad(src, wv, N, H, W, cup, absmax=amax, cskip=csk)
            assert relerr(o_up, ref) < BX3_TOL and relerr(o_sk, ref_sk) < BX3_TOL
        olds = rnd(N, csk, H, W, seed=10)
        o1 = dev(olds).clone()
        conv_sp_dgrad(src, wv, N, H, W, cup, absmax=amax, accum_into=dev(old).clone(), cskip=csk, skip_into=o1)
        assert relerr(o1, ref_sk + olds.double()) < BX3_TOL
    # any other channel counts: the skip channels' gradient as additional channel tiles of the launch (32 skip channels x 4 parities)
    if csk and not (cup <= 64 and csk <= 16):
        ref_sk = F.conv_transpose2d(dy.double(), w.double(), padding=1)[:, cup:]
        for batched in (False, True):
            wt = pack_spd(wd, cup, batched=batched, skip_tiles=True)
            assert bool(torch.isfinite(wt.view(torch.int16).float()).all())
            o_up, o_sk = conv_sp_dgrad(src, wt, N, H, W, cup, absmax=amax, cskip=csk)
            assert relerr(o_up, ref) < BX3_TOL and relerr(o_sk, ref_sk) < BX3_TOL
        olds = rnd(N, csk, H, W, seed=10)
        o1 = dev(olds).clone()
        conv_sp_dgrad(src, wt, N, H, W, cup, absmax=amax, accum_into=dev(old).clone(), cskip=csk, skip_into=o1)
        assert relerr(o1, ref_sk + olds.double()) < BX3_TOL


@pytest.mark.parametrize("cup,csk,cout,H,W,N", [(64, 16, 32, 16, 64, 2), (128, 24, 64, 24, 80, 2), (32, 0, 16, 20, 72, 3), (256, 32, 128, 8, 12, 2),
                                                 (40, 9, 24, 36, 70, 1), (72, 8, 300, 12, 20, 2), (1280, 96, 40, 4, 6, 4)])
def test_conv_sp_wgrad_matches_upsample_conv_autograd(hip, cup, csk, cout, H, W, N):
    """filter gradient of conv3x3(cat([nearest_up2(prev), skip])) for the up-sampled channels -- float64 autograd of the reference's op
    sequence -- against sc_conv3x3_sp_wgrad (nine plain GEMMs between the low-resolution source and tap-aligned 2x2 box sums of dy):
    BatchNorm / ReLU-backward operand with a range hint, BatchNorm + ReLU input, K not a multiple of the 32-pixel stage, row / column
    counts off the 256 x 128 tile, both column-block variants; the skip channels' columns through sc_wgrad_scatter_cols"""
    from starcop_amd._lib import TERMS_F16X2
    g, yraw = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    prev = rnd(N, cup, H // 2, W // 2, seed=3)
    sc0, sh0 = rnd(cup, seed=4) * 0.3 + 1, rnd(cup, seed=5) * 0.2
    cst = torch.zeros(cout, SC_CST)
    cst[:, 0], cst[:, 1] = rnd(cout, seed=6) * 0.3 + 1, rnd(cout, seed=7) * 0.2
    cst[:, 2], cst[:, 3], cst[:, 4] = rnd(cout, seed=8) * 0.5 + 1, rnd(cout, seed=9) * 0.1, rnd(cout, seed=10) * 0.05
    yh = yraw * cst[:, 0][None, :, None, None] + cst[:, 1][None, :, None, None]
    dy = torch.where(yh > 0, g, torch.zeros_like(g)) * cst[:, 2][None, :, None, None] + yraw * cst[:, 3][None, :, None, None] + cst[:, 4][None, :, None, None]
    up = F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2, mode="nearest").double()
    skip = rnd(N, max(csk, 1), H, W, seed=11)[:, :csk].double()
    xin = torch.cat([up, skip], 1).requires_grad_(False)
    w = torch.zeros(cout, cup + csk, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, padding=1).backward(dy.double())
    ref = w.grad
    amax = torch.tensor([float((cst[:, 2][None, :, None, None] * g).abs().max())], device=DEV)
    dys = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(yraw))
    src = make_src(dev(prev), cup, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))
    dw = wgrad_sp(dys, src, N, H, W, cout, cup + csk, absmax=amax)
    assert relerr(dw[:, :cup], ref[:, :cup]) < BX3_TOL
    if csk:
        assert bool(torch.isnan(dw[:, cup:]).all())            # the skip channels' columns are not this entry point's
        if csk % 8 == 0:
            dsk = wgrad_mfma(dys, [make_src(dev(skip.float()), csk, SRC_RAW)], N, H, W, cout, csk, 3, bx3=cout >= 32 and csk >= 32, terms=TERMS_F16X2 if (cout >= 32 and csk >= 32) else 0, absmax=amax)
            check(hip.sc_wgrad_scatter_cols(ptr(dsk), ptr(dw), cout, csk, cup + csk, cup, stream()))
            assert relerr(dw, ref) < BX3_TOL


@pytest.mark.parametrize("cin,cout,H,W,two", [(16, 64, 32, 32, False), (32, 16, 36, 70, True), (32, 32, 24, 40, False), (48, 40, 21, 32, False), (32, 64, 16, 32, True),
                                               (80, 32, 20, 36, True), (152, 64, 8, 64, True), (256, 256, 4, 6, False), (72, 136, 10, 33, False),
                                               (96, 32, 16, 32, False), (66, 20, 10, 38, True), (90, 32, 6, 64, False)])      # (65..96 -> <= 32: the 96-wide tile)
def test_conv_bx3_wgrad(hip, split_mode, cin, cout, H, W, two):
    """3x3 weight gradient with split-bf16 operands: vs fp64, and no worse than the fp32 MFMA kernel."""
    N = 3
    g, y = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6), rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where((yh > 0) & (yh < 6), g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    dys = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cst), aux=dev(y))
    if two:
        c0 = cin - 8
        prev, skip = rnd(N, c0, H // 2, W // 2, seed=11), rnd(N, 8, H, W, seed=12)
        sc0, sh0 = rnd(c0, seed=13) * 0.3 + 1, rnd(c0, seed=14) * 0.2
        xin = torch.cat([F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2), skip], 1)
        srcs = [make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0)), make_src(dev(skip), 8, SRC_RAW)]
    else:
        xin = rnd(N, cin, H, W, seed=11)
        srcs = [make_src(dev(xin), cin, SRC_RAW)]
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), w, padding=1).backward(dy.double())
    dw = wgrad_mfma(dys, srcs, N, H, W, cout, cin, 3, bx3=True)
    dw32 = wgrad_mfma(dys, srcs, N, H, W, cout, cin, 3)
    e, e32 = relerr(dw, w.grad), relerr(dw32, w.grad)
    assert e < 1e-5 and e < 3 * e32 + 2e-7, (e, e32)


@pytest.mark.parametrize("C_,H,W,stride", [(32, 32, 32, 1), (24, 20, 28, 2), (96, 16, 16, 2), (40, 7, 9, 1), (960, 2, 3, 1), (576, 4, 6, 2),
                                           (46, 16, 16, 1), (7, 16, 16, 1)])       # 16x16 stride 1: the wave-per-plane kernel (N*C % 4 == 0) and its fallback
def test_depthwise(hip, C_, H, W, stride):
    N = 2
    x, w = rnd(N, C_, H, W, seed=1), rnd(C_, 1, 3, 3, seed=2, scale=0.3)
    sc, sh = rnd(C_, seed=3) * 0.3 + 1, rnd(C_, seed=4) * 0.2
    xa = F.relu6(x * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xa, wr, stride=stride, padding=1, groups=C_)
    Ho, Wo = ref.shape[-2:]
    lib = hip
    src = make_src(dev(x), C_, SRC_AFFINE, act=ACT_RELU6, cst=cst_affine(sc, sh))
    out = torch.empty(N, C_, Ho, Wo, device=DEV)
    stats = torch.full((lib.sc_stat_rows(STAT_DW, N, Ho, Wo), C_, 2), float("nan"), device=DEV)
    wd = dev(w)
    check(lib.sc_dwconv3x3_fwd(C.byref(src), ptr(wd), ptr(out), N, C_, H, W, stride, ptr(stats), stream()))
    assert relerr(out, ref) < TOL
    assert relerr(stats.double().sum(0)[:, 0], ref.double().sum((0, 2, 3))) < 1e-4
    assert relerr(stats.double().sum(0)[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-4
    dy = rnd(N, C_, Ho, Wo, seed=5)
    ref.backward(dy)
    dys = make_src(dev(dy), C_, SRC_RAW)
    dx = torch.empty(N, C_, H, W, device=DEV)
    check(lib.sc_dwconv3x3_dgrad(C.byref(dys), ptr(wd), ptr(dx), 0, N, C_, H, W, stride, stream()))
    assert relerr(dx, xa.grad) < TOL
    acc = torch.zeros(C_ * 9, dtype=torch.float64, device=DEV)
    check(lib.sc_dwconv3x3_wgrad(C.byref(dys), C.byref(src), ptr(acc), N, C_, H, W, stride, stream()))
    assert relerr(acc.reshape(C_, 1, 3, 3), wr.grad) < TOL


@pytest.mark.parametrize("N,C_,H,W,stride", [(16, 96, 16, 16, 1), (3, 24, 32, 32, 1), (4, 40, 64, 64, 2), (2, 12, 64, 64, 1), (5, 7, 20, 28, 1), (16, 960, 16, 16, 1)])
def test_depthwise_finalizes_its_batchnorm_in_the_launch(hip, N, C_, H, W, stride):
    """sc_dwconv3x3_fwd_bn (producer-tail BatchNorm finalize: the last arrival of a channel, by ticket, sums that channel's statistics
    rows and writes constants / running statistics / the activation bound) against sc_dwconv3x3_fwd + sc_bn_finalize: identical
    outputs and rows, constants and running statistics equal to summation order, over several launches on the same tickets (the
    counters are monotone, never reset), bit-identical from launch to launch; wave-per-plane and tile kernels, one and two tiles per
    plane, ragged planes, stride 2"""
    from starcop_amd._lib import sc_bn_tail
    lib = hip
    x, w = rnd(N, C_, H, W, seed=1), rnd(C_, 1, 3, 3, seed=2, scale=0.3)
    sc, sh = rnd(C_, seed=3) * 0.3 + 1, rnd(C_, seed=4) * 0.2
    src = make_src(dev(x), C_, SRC_AFFINE, act=ACT_RELU6, cst=cst_affine(sc, sh))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    rows = lib.sc_stat_rows(STAT_DW, N, Ho, Wo)
    wd = dev(w)
    gamma, beta = dev(rnd(C_, seed=5) * 0.2 + 1), dev(rnd(C_, seed=6) * 0.1)
    res = {}
    for mode in ("separate", "tail"):
        out = torch.empty(N, C_, Ho, Wo, device=DEV)
        stats = torch.full((rows, C_, 2), float("nan"), device=DEV)
        rm, rv = torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV)
        cst = torch.full((C_, SC_CST), float("nan"), device=DEV)
        bound = torch.zeros(1, device=DEV)
        tickets = torch.zeros(C_, dtype=torch.int32, device=DEV)
        snaps = []
        for rep_ in range(3):
            if mode == "separate":
                check(lib.sc_dwconv3x3_fwd(C.byref(src), ptr(wd), ptr(out), N, C_, H, W, stride, ptr(stats), stream()))
                check(lib.sc_bn_finalize(ptr(stats), rows, float(N * Ho * Wo), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, 1, ptr(cst), C_,
                                         None, ptr(bound), stream()))
            else:
                bt = sc_bn_tail()
                bt.gamma, bt.beta, bt.running_mean, bt.running_var = gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr()
                bt.momentum, bt.eps, bt.cst, bt.act_bound, bt.tickets = 0.1, 1e-5, cst.data_ptr(), bound.data_ptr(), tickets.data_ptr()
                check(lib.sc_dwconv3x3_fwd_bn(C.byref(src), ptr(wd), ptr(out), N, C_, H, W, stride, ptr(stats), C.byref(bt), stream()))
            torch.cuda.synchronize()
            snaps.append(cst.clone())
        assert torch.equal(snaps[0], snaps[1]) and torch.equal(snaps[1], snaps[2])          # the constants do not depend on who arrives last
        if mode == "tail":
            assert bool((tickets == 3 * rows).all())
        res[mode] = (out.clone(), stats.clone(), cst.clone(), rm.clone(), rv.clone(), float(bound))
    a_, b_ = res["separate"], res["tail"]
    assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])
    assert bool(torch.isfinite(b_[2]).all())
    for k in (2, 3, 4):
        assert relerr(b_[k], a_[k]) < 1e-6, k
    assert abs(a_[5] - b_[5]) <= 1e-6 * a_[5]


@pytest.mark.parametrize("case", [(2, 4, 64, 96, "norm"), (1, 4, 130, 80, "raw"), (1, 3, 34, 200, "affine"), (2, 4, 17, 102, "norm"), (1, 1, 64, 64, "raw")],
                         ids=lambda c: "x".join(map(str, c)))
def test_stem_forward_shapes(hip, case):
    """the MFMA form of the <= 4-channel stem (k_stem_fwd4m: permuted pixels, 16-byte stores; taken when the output width is a multiple of
    4 and the output 16-byte aligned) and the VALU form (the same input through an output that is not 16-byte aligned) against
    F.conv2d(stride 2) of the prologue'd input in float64; statistics rows against the sums of the output"""
    N, Cin, H, W, mode = case
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x, w = rnd(N, Cin, H, W, seed=41) * 2.0, rnd(32, Cin, 3, 3, seed=42, scale=0.3)
    cst = torch.zeros(Cin, SC_CST)
    if mode == "norm":
        cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3] = 0.1, rnd(Cin, seed=43).abs() + 0.5, -1.0, 1.5
        xa = torch.clamp((x.double() - 0.1) / cst[:, 1].double()[None, :, None, None], -1.0, 1.5)
        src_of = lambda t: make_src(t, Cin, SRC_NORM, cst=dev(cst))
    elif mode == "affine":
        sc, sh = rnd(Cin, seed=44) * 0.3 + 1, rnd(Cin, seed=45) * 0.2
        xa = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
        src_of = lambda t: make_src(t, Cin, SRC_AFFINE, act=ACT_RELU, cst=cst_affine(sc, sh))
    else:
        xa = x.double()
        src_of = lambda t: make_src(t, Cin, SRC_RAW)
    ref = F.conv2d(xa, w.double(), stride=2, padding=1)
    xd, wd = dev(x), dev(w)
    rows = hip.sc_stat_rows(STAT_STEM, N, Ho, Wo)
    obuf = torch.zeros(N * 32 * Ho * Wo + 4, device=DEV)
    outs = []
    for off in (0, 1):
        out = obuf[off:off + N * 32 * Ho * Wo].view(N, 32, Ho, Wo)
        out.fill_(float("nan"))
        stats = torch.full((rows, 32, 2), float("nan"), device=DEV)
        check(hip.sc_stem_conv_fwd(C.byref(src_of(xd)), ptr(wd), ptr(out), N, Cin, H, W, ptr(stats), stream()))
        assert relerr(out, ref) < TOL, relerr(out, ref)
        st = stats.double().sum(0)
        assert relerr(st[:, 0], out.double().sum((0, 2, 3))) < 1e-5
        assert relerr(st[:, 1], (out.double() ** 2).sum((0, 2, 3))) < 1e-5
        outs.append(out.clone())
    assert relerr(outs[0], outs[1]) < 1e-5


def test_stem_fused_normalizer(hip):
    """stem conv reads raw products and applies clamp((x-off)/fac, lo, hi) on load (normalizer_module.py:134)."""
    N, Cin, H, W = 2, 4, 64, 96
    x = rnd(N, Cin, H, W, seed=1).abs() * torch.tensor([1500., 60., 60., 60.])[None, :, None, None]
    w = rnd(32, Cin, 3, 3, seed=2, scale=0.3)
    cst = torch.zeros(Cin, SC_CST)
    cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3] = 0., torch.tensor([1750., 60., 60., 60.]), 0., 2.
    xn = torch.clamp((x - 0) / cst[:, 1][None, :, None, None], 0, 2)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xn, wr, stride=2, padding=1)
    src = make_src(dev(x), Cin, SRC_NORM, cst=dev(cst))
    out = torch.empty(N, 32, H // 2, W // 2, device=DEV)
    stats = torch.full((hip.sc_stat_rows(STAT_STEM, N, H // 2, W // 2), 32, 2), float("nan"), device=DEV)
    check(hip.sc_stem_conv_fwd(C.byref(src), ptr(dev(w)), ptr(out), N, Cin, H, W, ptr(stats), stream()))
    assert relerr(out, ref) < TOL
    assert relerr(stats.double().sum(0)[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-4
    dy = rnd(N, 32, H // 2, W // 2, seed=3)
    ref.backward(dy)
    n = hip.sc_stem_wgrad_workspace_floats(N, Cin, H, W)
    ws, dw = torch.empty(n, device=DEV), torch.empty(32, Cin, 3, 3, device=DEV)
    dys = make_src(dev(dy), 32, SRC_RAW)
    check(hip.sc_stem_conv_wgrad(C.byref(dys), C.byref(src), ptr(ws), n, ptr(dw), N, Cin, H, W, stream()))
    assert relerr(dw, wr.grad) < TOL


# (N, Cin, Cout, H, W, source): the shapes the streaming pointwise forward (k_pw_stream) takes -- <= 32 input channels, up to 160 outputs
# (features.1's projection, the features.3 / .4 expansions) -- on planes >= 8192 pixels, with and without statistics; one long
# contraction (144 channels: stays on the LDS-staged kernel either way)
PWS_CASES = [(2, 32, 16, 128, 128, "affine6"), (1, 24, 24, 96, 96, "affine6"), (2, 24, 144, 96, 96, "raw"), (1, 24, 144, 128, 64, "affine"),
             (1, 32, 32, 128, 64, "affine6"), (1, 144, 32, 96, 96, "affine6"), (3, 32, 144, 64, 128, "affine"), (2, 32, 192, 64, 64, "raw"),
             (1, 24, 176, 64, 64, "affine")]


@pytest.mark.parametrize("case", PWS_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pointwise_streaming_forward(hip, case):
    """sc_conv2d_mfma, ks = 1, large planes: the 16-byte streaming kernel against the float64 convolution of the prologue'd source, its
    statistics rows against the sums of its own output, and against the LDS-staged kernel (which the entry point falls back to for a
    source that is not 16-byte aligned) on the same input"""
    N, Cin, Cout, H, W, mode = case
    x, w = rnd(N, Cin, H, W, seed=21) * 2.0, rnd(Cout, Cin, 1, 1, seed=22, scale=(2.0 / Cin) ** 0.5)
    sc, sh = rnd(Cin, seed=23) * 0.3 + 1, rnd(Cin, seed=24) * 0.5
    if mode == "raw":
        xa = x.double()
    else:
        xa = x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
        if mode == "affine6":
            xa = xa.clamp(0.0, 6.0)
    ref = F.conv2d(xa, w.double())
    co_t = 32 if Cout <= 32 else 64
    wpk = pack(dev(w), co_t, 0)
    buf = torch.zeros(x.numel() + 4, device=DEV)
    outs = []
    for off in (0, 1):
        xv = buf[off:off + x.numel()].view(N, Cin, H, W)
        xv.copy_(dev(x))
        src = (make_src(xv, Cin, SRC_RAW) if mode == "raw"
               else make_src(xv, Cin, SRC_AFFINE, act=ACT_RELU6 if mode == "affine6" else ACT_NONE, cst=cst_affine(sc, sh)))
        (out,), stats = conv_mfma([src], wpk, N, H, W, Cout, 1, co_t, want_stats=True)
        assert relerr(out, ref) < TOL, relerr(out, ref)
        assert bool(torch.isfinite(stats).all())
        st = stats.double().sum(0)
        assert relerr(st[:, 0], out.double().sum((0, 2, 3))) < 1e-5
        assert relerr(st[:, 1], (out.double() ** 2).sum((0, 2, 3))) < 1e-5
        (out2,), none = conv_mfma([src], wpk, N, H, W, Cout, 1, co_t, want_stats=False)
        assert none is None and torch.equal(out, out2)
        outs.append(out)
    assert relerr(outs[0], outs[1]) < 1e-6


# (N, layer Cin, layer Cout, H, W, act of the BatchNorm the gradient passes): the projections' data gradients the streaming kernel takes
# (few dy channels -> many outputs: features.1 / .2 / .3 at batch 16 are 16 -> 32 at 256^2, 24 -> 96 and 24 -> 144 at 128^2)
PWS_DGRAD_CASES = [(2, 32, 16, 128, 128, ACT_NONE), (1, 96, 24, 96, 96, ACT_NONE), (2, 144, 24, 128, 64, ACT_NONE), (1, 144, 24, 96, 96, ACT_RELU6),
                   (1, 160, 16, 64, 128, ACT_RELU), (2, 192, 32, 64, 64, ACT_NONE), (1, 144, 32, 64, 64, ACT_NONE), (1, 184, 24, 64, 64, ACT_RELU6)]


@pytest.mark.parametrize("case", PWS_DGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pointwise_streaming_dgrad(hip, case):
    """the same kernel with a BatchNorm-backward source (dy formed on load from (g, y)) against conv_transpose2d in float64, and against
    the LDS-staged kernel through sources that are not 16-byte aligned"""
    N, cin, cout, H, W, act = case
    g, y = rnd(N, cout, H, W, seed=31), rnd(N, cout, H, W, seed=32) * 2.0
    w = rnd(cout, cin, 1, 1, seed=33, scale=0.2)
    a, b = rnd(cout, seed=34) * 0.2 + 1, rnd(cout, seed=35) * 0.5
    A, B, D = rnd(cout, seed=36), rnd(cout, seed=37) * 0.1, rnd(cout, seed=38) * 0.1
    yh = (y * a[None, :, None, None] + b[None, :, None, None]).double()
    ok = torch.ones_like(yh, dtype=torch.bool) if act == ACT_NONE else ((yh > 0) if act == ACT_RELU else ((yh > 0) & (yh < 6)))
    dy = (torch.where(ok, g.double(), torch.zeros((), dtype=torch.float64)) * A.double()[None, :, None, None]
          + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None])
    ref = F.conv_transpose2d(dy, w.double())
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    co_t = 32 if cin <= 32 else 64
    wpk = pack(dev(w), co_t, 1)
    gb, yb = torch.zeros(g.numel() + 4, device=DEV), torch.zeros(g.numel() + 4, device=DEV)
    outs = []
    for off in (0, 1):
        gv, yv = gb[off:off + g.numel()].view_as(g), yb[off:off + g.numel()].view_as(g)
        gv.copy_(dev(g)); yv.copy_(dev(y))
        src = make_src(gv, cout, SRC_BNBWD, act=act, cst=dev(cst), aux=yv)
        (out,), _ = conv_mfma([src], wpk, N, H, W, cin, 1, co_t)
        assert relerr(out, ref) < TOL, relerr(out, ref)
        outs.append(out)
    assert relerr(outs[0], outs[1]) < 1e-6


@pytest.mark.parametrize("shape", [(1, 16, 64), (2, 48, 200), (1, 33, 70), (1, 17, 61), (2, 40, 1248 // 8)], ids=lambda s: "x".join(map(str, s)))
def test_head_forward_shapes(hip, shape):
    """the two staging forms of the Cin = 16 head forward (16-byte requests when W % 4 == 0 and the planes are 16-byte aligned, 4-byte
    otherwise) on whole / ragged tiles, and the 4-byte form on the SAME input through a source that is not 16-byte aligned"""
    N, H, W = shape
    Cin = 16
    x, w, b = rnd(N, Cin, H, W, seed=11), rnd(1, Cin, 3, 3, seed=12, scale=0.3), torch.tensor([-0.21])
    sc, sh = rnd(Cin, seed=13) * 0.3 + 1, rnd(Cin, seed=14) * 0.2
    ref = F.conv2d(F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), w, b, padding=1)
    wd, bd, cst = dev(w), dev(b), cst_affine(sc, sh)
    buf = torch.zeros(x.numel() + 4, device=DEV)
    outs = []
    for off in (0, 1):
        xv = buf[off:off + x.numel()].view(N, Cin, H, W)
        xv.copy_(dev(x))
        assert (xv.data_ptr() % 16 == 0) == (off == 0)
        src = make_src(xv, Cin, SRC_AFFINE, act=ACT_RELU, cst=cst)
        out = torch.full((N, 1, H, W), float("nan"), device=DEV)
        check(hip.sc_head_conv_fwd(C.byref(src), ptr(wd), ptr(bd), ptr(out), N, Cin, H, W, stream()))
        assert relerr(out, ref) < TOL
        outs.append(out)
    assert relerr(outs[0], outs[1]) < 1e-6


def test_head(hip):
    N, Cin, H, W = 2, 16, 40, 72
    x, w, b = rnd(N, Cin, H, W, seed=1), rnd(1, Cin, 3, 3, seed=2, scale=0.3), torch.tensor([0.37])
    sc, sh = rnd(Cin, seed=3) * 0.3 + 1, rnd(Cin, seed=4) * 0.2
    xa = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xa, wr, br, padding=1)
    src = make_src(dev(x), Cin, SRC_AFFINE, act=ACT_RELU, cst=cst_affine(sc, sh))
    out = torch.empty(N, 1, H, W, device=DEV)
    wd = dev(w)
    check(hip.sc_head_conv_fwd(C.byref(src), ptr(wd), ptr(dev(b)), ptr(out), N, Cin, H, W, stream()))
    assert relerr(out, ref) < TOL
    dl = rnd(N, 1, H, W, seed=5)
    ref.backward(dl)
    dld = dev(dl)
    gin = torch.empty(N, Cin, H, W, device=DEV)
    check(hip.sc_head_conv_dgrad(ptr(dld), ptr(wd), ptr(gin), N, Cin, H, W, stream()))
    assert relerr(gin, xa.grad) < TOL
    n = hip.sc_head_wgrad_workspace_floats(N, Cin, H, W)
    ws, dw, db = torch.empty(n, device=DEV), torch.empty(1, Cin, 3, 3, device=DEV), torch.empty(1, device=DEV)
    check(hip.sc_head_conv_wgrad(ptr(dld), C.byref(src), ptr(ws), n, ptr(dw), ptr(db), N, Cin, H, W, stream()))
    assert relerr(dw, wr.grad) < TOL and relerr(db, br.grad) < TOL
    # the one-sweep backward (Cin = 16): same gin bit for bit, same dW / dbias
    gin2 = torch.full((N, Cin, H, W), float("nan"), device=DEV)
    dw2, db2 = torch.empty(1, Cin, 3, 3, device=DEV), torch.empty(1, device=DEV)
    check(hip.sc_head_conv_bwd(ptr(dld), C.byref(src), ptr(wd), ptr(gin2), ptr(ws), n, ptr(dw2), ptr(db2), N, Cin, H, W, None, None, stream()))
    assert torch.equal(gin2, gin)
    assert relerr(dw2, wr.grad) < TOL and relerr(db2, br.grad) < TOL
    # ... and the BatchNorm-backward sums of its input in the same sweep == what sc_bn_bwd_reduce computes from (gin, x)
    cf = torch.zeros(Cin, SC_CST); cf[:, 0], cf[:, 1], cf[:, 2], cf[:, 3] = sc, sh, rnd(Cin, seed=6) * 0.1, rnd(Cin, seed=7).abs() + 0.5
    cfd = dev(cf)
    src2 = make_src(dev(x), Cin, SRC_AFFINE, act=ACT_RELU, cst=cfd)
    rows = hip.sc_head_bwd_bn_rows(N, H, W)
    sums = torch.full((rows, Cin, 2), float("nan"), dtype=torch.float64, device=DEV)
    amax = torch.zeros(1, device=DEV)
    gin3 = torch.empty(N, Cin, H, W, device=DEV)
    check(hip.sc_head_conv_bwd(ptr(dld), C.byref(src2), ptr(wd), ptr(gin3), ptr(ws), n, ptr(dw2), ptr(db2), N, Cin, H, W, ptr(sums), ptr(amax), stream()))
    assert torch.equal(gin3, gin) and not torch.isnan(sums).any()
    rrows = hip.sc_stat_rows(STAT_BNBWD, N, H, W)
    rsums = torch.empty(rrows * Cin * 2, dtype=torch.float64, device=DEV)
    ramax = torch.zeros(1, device=DEV)
    check(hip.sc_bn_bwd_reduce(ptr(gin), ptr(dev(x)), ptr(cfd), ACT_RELU, ptr(rsums), N, Cin, H * W, ptr(ramax), None, stream()))
    want = rsums.view(rrows, Cin, 2).sum(0)
    assert relerr(sums.sum(0)[:, 0], want[:, 0]) < 1e-6 and relerr(sums.sum(0)[:, 1], want[:, 1]) < 1e-6
    assert float(amax) == pytest.approx(float(ramax), rel=1e-6)


def test_batchnorm_bookkeeping(hip):
    """bn_finalize / bn_bwd_reduce / bn_bwd_finalize + the BNBWD prologue == autograd of F.batch_norm + relu6."""
    N, C_, H, W = 3, 24, 16, 20
    y = (rnd(N, C_, H, W, seed=1) * 2 + 1.5).requires_grad_(True)
    gamma, beta = (rnd(C_, seed=2) * 0.2 + 1).requires_grad_(True), (rnd(C_, seed=3) * 0.5 + 1.0).requires_grad_(True)
    rm, rv = torch.zeros(C_), torch.ones(C_)
    z = F.relu6(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    g = rnd(N, C_, H, W, seed=4)
    z.backward(g)
    yd = dev(y)
    stats = torch.zeros(300, C_, 2, device=DEV)           # partial rows: any split of the sums over rows
    stats[3, :, 0] = yd.double().sum((0, 2, 3)).float() - 5.0; stats[299, :, 0] = 5.0
    stats[7, :, 1] = (yd.double() ** 2).sum((0, 2, 3)).float()
    rmd, rvd = torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV)
    cst = torch.zeros(C_, SC_CST, device=DEV)
    cnt = float(N * H * W)
    check(hip.sc_bn_finalize(ptr(stats), 300, cnt, ptr(dev(gamma)), ptr(dev(beta)), ptr(rmd), ptr(rvd),
                             0.1, 1e-5, 1, ptr(cst), C_, None, None, stream()))
    assert relerr(rmd, rm) < 1e-5 and relerr(rvd, rv) < 1e-5
    # act_bound: the tensor's by-construction bound max_c |gamma_c| sqrt(count - 1) + |beta_c| (sticky, never lowered)
    slot = torch.tensor([3.0], device=DEV)
    check(hip.sc_bn_finalize(ptr(stats), 300, cnt, ptr(dev(gamma)), ptr(dev(beta)), ptr(rmd.clone()), ptr(rvd.clone()),
                             0.1, 1e-5, 1, ptr(cst.clone()), C_, None, ptr(slot), stream()))
    want_b = float((gamma.abs().double() * (cnt - 1) ** 0.5 + beta.abs().double()).max())
    assert want_b <= float(slot) <= want_b * (1 + 1e-5), (float(slot), want_b)
    assert float(F.batch_norm(y.detach(), None, None, gamma.detach(), beta.detach(), True).abs().max()) <= float(slot)      # (and far below it)
    # many rows (full-resolution layers): the coalesced pre-reduction into 64 fp64 partial rows gives the same constants
    nr = 5000
    big = torch.zeros(nr, C_, 2, device=DEV)
    big[:, :, 0] = (yd.double().sum((0, 2, 3)) / nr).float()[None, :] + torch.randn(nr, C_, device=DEV) * 1e-3
    big[:, :, 1] = ((yd.double() ** 2).sum((0, 2, 3)) / nr).float()[None, :]
    outs = []
    for scr in (None, torch.empty(64 * 2 * C_, dtype=torch.float64, device=DEV)):
        rm2, rv2, c2 = torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV), torch.zeros(C_, SC_CST, device=DEV)
        check(hip.sc_bn_finalize(ptr(big), nr, cnt, ptr(dev(gamma)), ptr(dev(beta)), ptr(rm2), ptr(rv2), 0.1, 1e-5, 1, ptr(c2), C_, ptr(scr), None, stream()))
        outs.append((rm2, rv2, c2))
    assert all(relerr(a, b) < 1e-6 for a, b in zip(outs[0], outs[1]))
    m_want = big[:, :, 0].double().sum(0) / cnt
    assert relerr(outs[1][2][:, 2], m_want) < 1e-6
    src = make_src(yd, C_, SRC_AFFINE, act=ACT_RELU6, cst=cst)
    zz = torch.empty_like(yd)
    check(hip.sc_apply_src(C.byref(src), ptr(zz), N, C_, H * W, stream()))
    assert relerr(zz, z) < 1e-5
    nrows = hip.sc_stat_rows(STAT_BNBWD, N, H, W)
    sums = torch.full((nrows, C_, 2), float("nan"), dtype=torch.float64, device=DEV)
    gd = dev(g)
    check(hip.sc_bn_bwd_reduce(ptr(gd), ptr(yd), ptr(cst), ACT_RELU6, ptr(sums), N, C_, H * W, None, None, stream()))
    dgm, dbt, cstb = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV), torch.zeros(C_, SC_CST, device=DEV)
    check(hip.sc_bn_bwd_finalize(ptr(sums), nrows, cnt, ptr(cst), ptr(dgm), ptr(dbt), ptr(cstb), C_, stream()))
    assert relerr(dgm, gamma.grad) < TOL and relerr(dbt, beta.grad) < TOL
    # one-launch form for few-pixel layers
    dg2, db2, cb2 = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV), torch.full((C_, SC_CST), float("nan"), device=DEV)
    check(hip.sc_bn_bwd_small(ptr(gd), ptr(yd), ptr(cst), ACT_RELU6, N, C_, H * W, ptr(dg2), ptr(db2), ptr(cb2), None, None, stream()))
    assert relerr(dg2, gamma.grad) < TOL and relerr(db2, beta.grad) < TOL and relerr(cb2, cstb) < 1e-5
    dsrc = make_src(gd, C_, SRC_BNBWD, act=ACT_RELU6, cst=cstb, aux=yd)
    dyd = torch.empty_like(yd)
    check(hip.sc_apply_src(C.byref(dsrc), ptr(dyd), N, C_, H * W, stream()))
    assert relerr(dyd, y.grad) < TOL


def test_downsum_and_add(hip):
    N, C_, H, W = 2, 12, 8, 10
    x = rnd(N, C_, 2 * H, 2 * W, seed=1)
    ref = F.avg_pool2d(x, 2) * 4
    out = torch.empty(N, C_, H, W, device=DEV)
    check(hip.sc_downsum2x2(ptr(dev(x)), ptr(out), 0, N, C_, H, W, stream()))
    assert relerr(out, ref) < 1e-6
    a, b = rnd(N, C_, H, W, seed=2), rnd(N, C_, H, W, seed=3)
    sc, sh = rnd(C_, seed=4), rnd(C_, seed=5)
    sa = make_src(dev(a), C_, SRC_RAW)
    sb = make_src(dev(b), C_, SRC_AFFINE, act=ACT_NONE, cst=cst_affine(sc, sh))
    o2 = torch.empty(N, C_, H, W, device=DEV)
    check(hip.sc_add_srcs(C.byref(sa), C.byref(sb), ptr(o2), N, C_, H * W, stream()))
    assert relerr(o2, a + b * sc[None, :, None, None] + sh[None, :, None, None]) < 1e-6


@pytest.mark.parametrize("N,C_,H,W", [(2, 24, 48, 64), (1, 5, 45, 44), (2, 7, 9, 11), (1, 3, 64, 80)])
def test_add_srcs_forms_and_range_record(hip, N, C_, H, W):
    """sc_add_srcs_absmax: the 16-byte form (H*W % 4 == 0, aligned tensors; chunks whose second half is empty or ragged) and the 4-byte form
    (the same operands through views that are not 16-byte aligned), RAW + AFFINE and BNBWD + RAW operands, with the max |sum| record"""
    a, b = rnd(N, C_, H, W, seed=71), rnd(N, C_, H, W, seed=72)
    sc, sh = rnd(C_, seed=73) * 0.3 + 1, rnd(C_, seed=74) * 0.2
    y = rnd(N, C_, H, W, seed=75)
    cb = torch.zeros(C_, SC_CST); cb[:, 0], cb[:, 1], cb[:, 2], cb[:, 3], cb[:, 4] = sc, sh, rnd(C_, seed=76), rnd(C_, seed=77) * 0.1, rnd(C_, seed=78) * 0.1
    want1 = a.double() + F.relu6(b.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    yh = y.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    dy = (torch.where((yh > 0) & (yh < 6), a.double(), torch.zeros((), dtype=torch.float64)) * cb[:, 2].double()[None, :, None, None]
          + cb[:, 3].double()[None, :, None, None] * y.double() + cb[:, 4].double()[None, :, None, None])
    want2 = dy + b.double()
    n = a.numel()
    bufs = [torch.zeros(n + 4, device=DEV) for _ in range(4)]
    for off in (0, 1):
        av, bv, yv, ov = (t[off:off + n].view(N, C_, H, W) for t in bufs)
        av.copy_(dev(a)); bv.copy_(dev(b)); yv.copy_(dev(y))
        for want, sa, sb in ((want1, make_src(av, C_, SRC_RAW), make_src(bv, C_, SRC_AFFINE, act=ACT_RELU6, cst=cst_affine(sc, sh))),
                             (want2, make_src(av, C_, SRC_BNBWD, act=ACT_RELU6, cst=dev(cb), aux=yv), make_src(bv, C_, SRC_RAW))):
            ov.fill_(float("nan"))
            amax = torch.zeros(1, device=DEV)
            check(hip.sc_add_srcs_absmax(C.byref(sa), C.byref(sb), ptr(ov), N, C_, H * W, ptr(amax), stream()))
            assert relerr(ov, want) < 1e-6
            assert abs(float(amax) - float(want.abs().max())) <= 1e-5 * float(want.abs().max())
    ad = dev(a)
    src1 = make_src(ad, C_, SRC_RAW)
    o1 = torch.empty(N, C_, H, W, device=DEV)
    check(hip.sc_apply_src(C.byref(src1), ptr(o1), N, C_, H * W, stream()))       # (one operand)
    assert torch.equal(o1, ad)


def test_bce_loss_and_grad(hip):
    n = 5000
    z, t, w = (rnd(n, seed=1) * 3).requires_grad_(True), (rnd(n, seed=2) > 0.5).float(), rnd(n, seed=3).abs().clamp(0.1, 1)
    for pw in (1.0, 15.0):
        z.grad = None
        l = F.binary_cross_entropy_with_logits(z, t, pos_weight=torch.tensor(pw), reduction="none")
        (l * w).mean().backward()
        acc = torch.zeros(1, dtype=torch.float64, device=DEV)
        dz, px = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        check(hip.sc_bce_logits_weighted(ptr(dev(z)), ptr(dev(t)), ptr(dev(w)), pw, n, ptr(acc), ptr(dz), ptr(px), stream()))
        assert relerr(px, l) < 1e-5
        assert abs(float(acc) / n - float((l * w).mean())) < 1e-5 * max(1.0, float((l * w).mean()))
        assert relerr(dz, z.grad) < 1e-5


def test_adam_matches_torch(hip):
    n = 4097
    p0, grads = rnd(n, seed=1), [rnd(n, seed=10 + i) for i in range(4)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-2)
    p, m, v = dev(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step, lr_d, hp = torch.zeros(1, dtype=torch.int64, device=DEV), torch.full((1,), 1e-2, device=DEV), torch.zeros(4, device=DEV)
    for g in grads:
        pr.grad = g.clone(); opt.step()
        check(hip.sc_adam_prepare(ptr(step), ptr(lr_d), 0.9, 0.999, ptr(hp), stream()))
        check(hip.sc_adam_step(ptr(p), ptr(dev(g)), ptr(m), ptr(v), n, 0.0, 0.9, 0.999, 1e-8, 0.0, 1.0, 1.0, 1.0, ptr(hp), stream()))
    assert int(step) == 4
    assert relerr(p, pr) < 1e-5


def test_masks_bit_exact(hip):
    """pred_binary / differences / pred_classification are integer outputs: bit-exact incl. ties (logit == 0,
    count == threshold)."""
    B, H, W = 3, 64, 64        # threshold = 10*64*64/64^2 = 10 pixels
    z = rnd(B, 1, H, W, seed=1)
    z[0] = -1.0; z[0, 0, 0, :10] = 1.0            # exactly 10 positives  -> NOT a plume tile (strict >)
    z[1] = -1.0; z[1, 0, 0, :11] = 1.0            # 11 positives -> plume tile
    z[2, 0, 0, :5] = 0.0                          # logit == 0: ">= 0" true, "sigmoid > .5" false
    t = (rnd(B, 1, H, W, seed=2) > 0.3).float()
    for ge0 in (0, 1):
        pred = torch.empty(B, 1, H, W, device=DEV)
        pb = torch.empty(B, 1, H, W, dtype=torch.int64, device=DEV)
        df = torch.empty(B, 1, H, W, dtype=torch.int64, device=DEV)
        cnt, cls = torch.zeros(B, dtype=torch.int64, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
        check(hip.sc_threshold_masks(ptr(dev(z)), ptr(dev(t)), ge0, ptr(pred), ptr(pb), ptr(df), ptr(cnt), B, H * W, stream()))
        check(hip.sc_pred_classification(ptr(cnt), ptr(cls), B, H, W, stream()))
        rb = (z >= 0).long() if ge0 else (torch.sigmoid(z) > .5).long()
        assert torch.equal(pb.cpu(), rb)
        assert torch.equal(df.cpu(), 2 * rb + (t.long() == 1).long())
        assert torch.equal(cls.cpu(), (rb.sum((-1, -2)) > 10 * H * W / 64 ** 2).long().reshape(-1))
        assert relerr(pred, torch.sigmoid(z)) < 1e-6
    assert cls.cpu().tolist()[:2] == [0, 1]


@pytest.mark.parametrize("gscale", [1.0, 1e-6])
@pytest.mark.parametrize("cin,cout,H,W,up", [(16, 16, 36, 70, 0), (32, 16, 24, 40, 1), (16, 9, 8, 33, 0), (32, 16, 64, 64, 0), (32, 16, 64, 96, 1), (32, 11, 10, 34, 1)])
def test_conv_thin16_two_fp16_terms(hip, cin, cout, H, W, up, gscale):
    """sc_conv3x3_thin16 (decoder.blocks.4: filters in registers, one staging pass, v_mfma_f32_16x16x32_f16): forward with
    BatchNorm/ReLU prologue (+ nearest x2 upsampling) and statistics rows, and backward-data from transposed filters with the
    BatchNorm-backward prologue, against fp64."""
    from starcop_amd._lib import TERMS_F16X2, sc_conv_args
    import ctypes as C
    import hip_ops
    N = 2
    lib = hip

    def pack_thin(w, tflip):
        co, ci = w.shape[:2]
        out = torch.empty(lib.sc_packed_weight_floats_thin16(co, ci, tflip), device=DEV)
        check(lib.sc_pack_weights_thin16(ptr(w), ptr(out), co, ci, tflip, stream()))
        return out

    def run(src, wpk, Cout_, want_stats=False, absmax=None, bnr=None):
        a = sc_conv_args()
        a.nsrc = 1; a.src[0] = src
        a.wpk = wpk.data_ptr()
        a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, Cout_, 3, 16
        out = torch.full((N, Cout_, H, W), float("nan"), device=DEV)
        a.out0 = out.data_ptr(); a.csplit = Cout_; a.terms = TERMS_F16X2
        rows = lib.sc_stat_rows(_lib.STAT_CONV3, N, H, W)
        st = torch.full((rows, Cout_, 2), float("nan"), device=DEV) if want_stats else None
        a.stats = st.data_ptr() if want_stats else None
        a.absmax = absmax.data_ptr() if absmax is not None else None
        if bnr is not None:
            a.bnr = C.addressof(hip_ops.make_bnr(bnr, rows, Cout_))
        check(lib.sc_conv3x3_thin16(C.byref(a), stream()))
        return out, st

    x = rnd(N, cin, H >> up, W >> up, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.2)
    sc, sh = rnd(cin, seed=3) * 0.3 + 1.0, rnd(cin, seed=4) * 0.2
    xin = torch.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.double(), padding=1)
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, up=up, cst=cst_affine(sc, sh))
    out, st = run(src, pack_thin(dev(w), 0), cout, want_stats=True)
    assert relerr(out, ref) < 2e-6
    assert relerr(st.sum(0)[:, 0], ref.sum((0, 2, 3))) < 1e-5 and relerr(st.sum(0)[:, 1], (ref * ref).sum((0, 2, 3))) < 1e-5
    if up or cout != 16:
        return
    # backward-data: gradient of the cin-channel input from (g, y) of the cout-channel output; here cout == 16 so that the
    # transposed problem (rows = cin <= 16 or reduction = 16) is a thin one when cin == 16
    if cin != 16:
        return
    g, y = rnd(N, cout, H, W, seed=5) * gscale, rnd(N, cout, H, W, seed=6)
    a_, b_ = rnd(cout, seed=7) * 0.2 + 1, rnd(cout, seed=8) * 0.2
    A, B, D = a_.clone(), rnd(cout, seed=9) * 0.1 * gscale, rnd(cout, seed=10) * 0.1 * gscale
    yh = y * a_[None, :, None, None] + b_[None, :, None, None]
    gm = torch.where(yh > 0, g, torch.zeros(()))
    dy = gm.double() * A.double()[None, :, None, None] + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None]
    cstb = torch.zeros(cout, SC_CST); cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = a_, b_, A, B, D
    amax = torch.tensor([float((gm.abs().amax((0, 2, 3)) * a_.abs()).max())], device=DEV)
    dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cstb), aux=dev(y))
    dx, _ = run(dsrc, pack_thin(dev(w), 1), cin, absmax=amax)
    assert relerr(dx, F.conv_transpose2d(dy, w.double(), padding=1)) < 1e-5
    # ... and with the BatchNorm-backward sums of the tensor that gradient belongs to (sc_bnr_args)
    import hip_ops
    y_in = rnd(N, cin, H, W, seed=11)
    cst_in = torch.zeros(cin, SC_CST)
    cst_in[:, 0], cst_in[:, 1], cst_in[:, 2], cst_in[:, 3] = rnd(cin, seed=12) * 0.3 + 1, rnd(cin, seed=13) * 0.3, rnd(cin, seed=14) * 0.2, rnd(cin, seed=15).abs() + 0.5
    dx2, _ = run(dsrc, pack_thin(dev(w), 1), cin, absmax=amax, bnr=(dev(y_in), dev(cst_in), ACT_RELU))
    assert torch.equal(dx2, dx)
    _bnr_check(lib, dx2, y_in, cst_in, ACT_RELU, N, H * W)


@pytest.mark.parametrize("gscale", [1.0, 1e-6])
@pytest.mark.parametrize("N,H,W,accum", [(2, 64, 128, 0), (1, 24, 40, 1), (2, 10, 34, 0), (1, 128, 64, 0)])
def test_conv_thin16_half_resolution_data_gradient(hip, N, H, W, accum, gscale):
    """sc_conv3x3_thin16 with down0 (k_conv3_thin_spd): the data gradient of a 32 -> 16 channel 3x3 layer whose source was up-sampled 2x,
    stored 2x2-summed at half resolution, from (g, y) of the 16-channel output -- against conv_transpose2d + the 2x2 sum in float64
    (whole and ragged tiles; accumulate on / off; tiny gradients through the range hint)"""
    from starcop_amd._lib import TERMS_F16X2, sc_conv_args
    cin, cout = 32, 16
    w = rnd(cout, cin, 3, 3, seed=52, scale=0.2)
    g, y = rnd(N, cout, H, W, seed=55) * gscale, rnd(N, cout, H, W, seed=56)
    a_, b_ = rnd(cout, seed=57) * 0.2 + 1, rnd(cout, seed=58) * 0.2
    A, B, D = a_.clone(), rnd(cout, seed=59) * 0.1 * gscale, rnd(cout, seed=60) * 0.1 * gscale
    yh = y * a_[None, :, None, None] + b_[None, :, None, None]
    gm = torch.where(yh > 0, g, torch.zeros(()))
    dy = gm.double() * A.double()[None, :, None, None] + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None]
    ref = F.conv_transpose2d(dy, w.double(), padding=1)
    ref = ref.reshape(N, cin, H // 2, 2, W // 2, 2).sum((3, 5))
    cstb = torch.zeros(cout, SC_CST); cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = a_, b_, A, B, D
    amax = torch.tensor([float((gm.abs().amax((0, 2, 3)) * a_.abs()).max())], device=DEV)
    wpk = torch.empty(hip.sc_packed_weight_floats_thin16(cout, cin, 1), device=DEV)
    check(hip.sc_pack_weights_thin16(ptr(dev(w)), ptr(wpk), cout, cin, 1, stream()))
    old = rnd(N, cin, H // 2, W // 2, seed=61) * gscale
    out = dev(old).clone() if accum else torch.full((N, cin, H // 2, W // 2), float("nan"), device=DEV)
    a = sc_conv_args()
    a.nsrc = 1; a.src[0] = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cstb), aux=dev(y))
    a.wpk = wpk.data_ptr()
    a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, cin, 3, 16
    a.out0 = out.data_ptr(); a.csplit = cin; a.terms = TERMS_F16X2; a.down0 = 1; a.accum0 = accum
    a.absmax = amax.data_ptr()
    check(hip.sc_conv3x3_thin16(C.byref(a), stream()))
    assert relerr(out, ref + (old.double() if accum else 0.0)) < 1e-5


def _bnr_reference(dx, y_in, cst_in, act):
    """what sc_bn_bwd_reduce computes from a gradient dx of a BatchNorm'd tensor with raw values y_in: (sum g', sum g' x_hat, max |scale g'|)"""
    sc, sh, mu, isd = (cst_in[:, k].double()[None, :, None, None] for k in range(4))
    yh = y_in.double() * sc + sh
    m = (yh > 0) if act == ACT_RELU else ((yh > 0) & (yh < 6))
    gp = torch.where(m, dx.double().cpu(), torch.zeros((), dtype=torch.float64))
    return gp.sum((0, 2, 3)), (gp * (y_in.double() - mu) * isd).sum((0, 2, 3)), float((gp * sc).abs().max())


def _bnr_check(lib, dx, y_in, cst_in, act, N, HW):
    """the rows / range hint a data-gradient launch left (hip_ops.LAST_BNR) against the float64 sums, and -- through
    sc_bn_bwd_finalize_rows32 -- against the separate pass sc_bn_bwd_reduce + sc_bn_bwd_finalize over the same (dx, y)"""
    import hip_ops
    rows, amax = hip_ops.LAST_BNR
    C_ = y_in.shape[1]
    s1, s2, mx = _bnr_reference(dx, y_in, cst_in, act)
    assert bool(torch.isfinite(rows).all())
    got = rows.double().sum(0).cpu()
    scale = max(float(s1.abs().max()), float(s2.abs().max()), 1e-3)
    assert float((got[:, 0] - s1).abs().max()) < 2e-5 * scale and float((got[:, 1] - s2).abs().max()) < 2e-5 * scale
    assert abs(float(amax) - mx) <= 1e-6 * mx
    out = {}
    for name in ("fused", "separate"):
        dg, db, cb = (torch.full((n_,), float("nan"), device=DEV) for n_ in (C_, C_, C_ * SC_CST))
        if name == "fused":
            check(lib.sc_bn_bwd_finalize_rows32(ptr(rows), rows.shape[0], float(N * HW), ptr(dev(cst_in)), ptr(dg), ptr(db), ptr(cb), C_,
                                                ptr(torch.empty(64 * 2 * C_, dtype=torch.float64, device=DEV)) if rows.shape[0] >= 4096 else None, stream()))
        else:
            nr = lib.sc_stat_rows(_lib.STAT_BNBWD, N, dx.shape[2], dx.shape[3])
            sums = torch.empty(nr * C_ * 2, dtype=torch.float64, device=DEV)
            check(lib.sc_bn_bwd_reduce(ptr(dx), ptr(dev(y_in)), ptr(dev(cst_in)), act, ptr(sums), N, C_, HW, None, None, stream()))
            check(lib.sc_bn_bwd_finalize(ptr(sums), nr, float(N * HW), ptr(dev(cst_in)), ptr(dg), ptr(db), ptr(cb), C_, stream()))
        out[name] = (dg.cpu(), db.cpu(), cb.cpu())
    for a_, b_ in zip(out["fused"], out["separate"]):
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-3)


@pytest.mark.parametrize("cin,cout,co_t,H,W,cs,down0", [(64, 32, 64, 20, 40, None, False), (40, 24, 32, 36, 70, None, False), (80, 32, 32, 16, 64, 64, False),
                                                     (96, 16, 64, 24, 48, 64, True), (32, 16, 32, 34, 66, None, True), (152, 64, 64, 12, 32, 128, False)])
def test_conv_bx3_dgrad_leaves_batchnorm_backward_sums(hip, cin, cout, co_t, H, W, cs, down0):
    """sc_bnr_args on sc_conv3x3_bx3: the data-gradient launch that writes a single-consumer tensor's complete gradient also leaves
    that tensor's BatchNorm-backward sums and range hint -- plain and 2x2-down-summed out0, with and without a second (skip) output,
    ragged tiles; the gradient itself is bit-identical to the launch without the sums"""
    from starcop_amd._lib import TERMS_F16X2
    N = 2
    g, y = rnd(N, cout, H, W, seed=1), rnd(N, cout, H, W, seed=2)
    w = rnd(cout, cin, 3, 3, seed=3, scale=0.2)
    cst = torch.zeros(cout, SC_CST)
    cst[:, 0], cst[:, 1] = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    cst[:, 2], cst[:, 3], cst[:, 4] = rnd(cout, seed=6) * 0.5 + 1, rnd(cout, seed=7) * 0.1, rnd(cout, seed=8) * 0.1
    src = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(y))
    c0 = cin if cs is None else cs                      # channels of out0 = of the tensor whose sums are wanted
    Hq, Wq = (H // 2, W // 2) if down0 else (H, W)
    y_in = rnd(N, c0, Hq, Wq, seed=11)
    cst_in = torch.zeros(c0, SC_CST)
    cst_in[:, 0], cst_in[:, 1] = rnd(c0, seed=12) * 0.3 + 1, rnd(c0, seed=13) * 0.3
    cst_in[:, 2], cst_in[:, 3] = rnd(c0, seed=14) * 0.2, rnd(c0, seed=15).abs() + 0.5
    wpk = pack_bx3(dev(w), co_t, 1, TERMS_F16X2)
    amax = torch.tensor([float((cst[:, 2][None, :, None, None] * g).abs().max())], device=DEV)
    kw = dict(bx3=True, terms=TERMS_F16X2, csplit=cs, down0=down0, absmax=amax)
    plain, _ = conv_mfma([src], wpk, N, H, W, cin, 3, co_t, **kw)
    for act in (ACT_RELU, ACT_RELU6):
        outs, _ = conv_mfma([src], wpk, N, H, W, cin, 3, co_t, bnr=(dev(y_in), dev(cst_in), act), **kw)
        for a_, b_ in zip(outs, plain):
            assert torch.equal(a_, b_)
        _bnr_check(hip, outs[0], y_in, cst_in, act, N, Hq * Wq)


@pytest.mark.parametrize("stride,C_,H,W", [(1, 8, 64, 96), (2, 8, 64, 96), (1, 24, 32, 32), (2, 24, 32, 64), (1, 40, 16, 16), (2, 40, 16, 16),
                                           (1, 5, 70, 50), (2, 6, 36, 20), (1, 6, 16, 16), (1, 5, 16, 16)])
def test_dw_bwd_fused(hip, stride, C_, H, W):
    """depthwise backward in one pass (sc_dwconv3x3_bwd_fused): dx, dW and the BatchNorm-backward sums of the input, against
    autograd on the same graph  x -> BN affine + ReLU6 -> depthwise conv -> y ; dy = BNBWD(g, y)"""
    lib = _lib.load()
    N = 2
    x = rnd(N, C_, H, W, seed=1, scale=2.0)
    w = rnd(C_, 1, 3, 3, seed=2, scale=0.4)
    sc, sh = rnd(C_, seed=3) * 0.3 + 1.0, rnd(C_, seed=4) * 0.5 + 1.0
    mean, invstd = rnd(C_, seed=5) * 0.2, rnd(C_, seed=6).abs() + 0.5
    xa = x.clone().requires_grad_(True)
    pre = xa * sc[None, :, None, None] + sh[None, :, None, None]
    act = F.relu6(pre)
    wq = w.clone().requires_grad_(True)
    y = F.conv2d(act, wq, stride=stride, padding=1, groups=C_)
    Ho, Wo = y.shape[-2:]
    g = rnd(N, C_, Ho, Wo, seed=7)
    a, b = rnd(C_, seed=8) * 0.2 + 1, rnd(C_, seed=9) * 0.2
    A, B, D = rnd(C_, seed=10), rnd(C_, seed=11) * 0.1, rnd(C_, seed=12) * 0.1
    yd = y.detach()
    yh = yd * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where((yh > 0) & (yh < 6), g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * yd + D[None, :, None, None]
    y.backward(dy)
    dact = torch.autograd.grad(F.conv2d(act.detach().requires_grad_(True), w, stride=stride, padding=1, groups=C_), [], allow_unused=True) if False else None
    # dx of the kernel = gradient w.r.t. the ACTIVATED input (the consumer-side prologue backward is applied later through BNBWD)
    act2 = act.detach().clone().requires_grad_(True)
    F.conv2d(act2, w, stride=stride, padding=1, groups=C_).backward(dy)
    dx_ref = act2.grad
    pre_d = pre.detach()
    gbn = torch.where((pre_d > 0) & (pre_d < 6), dx_ref, torch.zeros(()))
    s1_ref = gbn.double().sum((0, 2, 3))
    s2_ref = (gbn.double() * ((x.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None])).sum((0, 2, 3))
    cst_d = torch.zeros(C_, SC_CST); cst_d[:, 0], cst_d[:, 1], cst_d[:, 2], cst_d[:, 3], cst_d[:, 4] = a, b, A, B, D
    cst_x = torch.zeros(C_, SC_CST); cst_x[:, 0], cst_x[:, 1], cst_x[:, 2], cst_x[:, 3] = sc, sh, mean, invstd
    dsrc = make_src(dev(g), C_, SRC_BNBWD, act=ACT_RELU6, cst=dev(cst_d), aux=dev(yd))
    xsrc = make_src(dev(x), C_, SRC_AFFINE, act=ACT_RELU6, cst=dev(cst_x))
    dx = torch.full((N, C_, H, W), float("nan"), device=DEV)
    acc = torch.zeros(C_ * 9, dtype=torch.float64, device=DEV)
    rows = lib.sc_stat_rows(STAT_DW, N, H, W)
    sums = torch.full((rows, C_, 2), float("nan"), dtype=torch.float64, device=DEV)
    check(lib.sc_dwconv3x3_bwd_fused(C.byref(dsrc), C.byref(xsrc), ptr(dev(w)), ptr(dx), ptr(acc), ptr(sums), N, C_, H, W, stride, stream()))
    assert relerr(dx, dx_ref) < TOL
    assert relerr(acc.reshape(C_, 1, 3, 3), wq.grad) < TOL
    tot = sums.sum(0).cpu()
    assert relerr(tot[:, 0], s1_ref) < TOL and relerr(tot[:, 1], s2_ref) < TOL
    # and it equals the three separate kernels it replaces
    dx2 = torch.empty_like(dx)
    check(lib.sc_dwconv3x3_dgrad(C.byref(dsrc), ptr(dev(w)), ptr(dx2), 0, N, C_, H, W, stride, stream()))
    assert torch.equal(dx, dx2)


def test_wgrad_deferred_batch_reduce(hip):
    """pointwise weight gradients with the reduction of their K-slice partials deferred into ONE batched launch
    (sc_conv2d_wgrad_mfma_deferred + sc_wgrad_reduce_batch) == autograd, for several layers of different shape at once"""
    import numpy as np
    from starcop_amd._lib import sc_wgrad_args, sc_wgrad_pending
    lib = _lib.load()
    N = 2
    cases = [(96, 16, 32, 32), (24, 144, 16, 24), (160, 960, 4, 4), (1280, 320, 2, 3), (8, 40, 9, 7)]
    pend, wants, outs, keep = [], [], [], []
    for k, (co, ci, H, W) in enumerate(cases):
        x, g = rnd(N, ci, H, W, seed=10 + k), rnd(N, co, H, W, seed=20 + k)
        wants.append(torch.einsum("nohw,nihw->oi", g, x).reshape(co, ci, 1, 1))
        a = sc_wgrad_args()
        xd, gd = dev(x), dev(g)
        a.dy = make_src(gd, co, SRC_RAW)
        a.nsrc = 1
        a.src[0] = make_src(xd, ci, SRC_RAW)
        a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, co, ci, 1
        nfl = lib.sc_wgrad_workspace_floats(N, H, W, co, ci, 1)
        part = torch.empty(nfl, device=DEV); dw = torch.full((co, ci, 1, 1), float("nan"), device=DEV)
        a.part, a.part_floats, a.dw = part.data_ptr(), nfl, dw.data_ptr()
        p = sc_wgrad_pending()
        check(lib.sc_conv2d_wgrad_mfma_deferred(C.byref(a), C.byref(p), stream()))
        assert p.total == co * ci and p.nparts >= 1
        pend.append(p); outs.append(dw); keep += [part, xd, gd]
    starts, nblk = [], 0
    for p in pend:
        starts.append(nblk); nblk += -(-int(p.total) // 256)
    descs = torch.from_numpy(np.frombuffer(b"".join(bytes(p) for p in pend), dtype=np.uint8).copy()).to(DEV)
    st_d = torch.tensor(starts, dtype=torch.int32, device=DEV)
    check(lib.sc_wgrad_reduce_batch(ptr(descs), ptr(st_d), len(pend), nblk, stream()))
    for dw, want in zip(outs, wants):
        assert relerr(dw, want) < TOL


@pytest.mark.parametrize("cin,cout,H,W,up,bnb", [(16, 16, 32, 64, 0, True), (32, 16, 16, 32, 0, True), (32, 16, 24, 40, 1, True), (16, 8, 20, 36, 0, False),
                                                   (32, 16, 64, 64, 1, True), (16, 16, 9, 34, 0, True), (16, 16, 3, 70, 0, True), (32, 12, 5, 32, 0, False)])
def test_wgrad_thin16_split(hip, cin, cout, H, W, up, bnb):
    """thin 3x3 weight gradient on the two-fp16-term 16x16x32 MFMA (sc_conv3x3_wgrad_thin16: decoder.blocks.4) against autograd
    in fp64: BNBWD / affine dy prologues, affine + ReLU input prologue, nearest-x2 upsampled input, ragged edges, range hint"""
    from starcop_amd._lib import sc_wgrad_args, TERMS_F16X2
    lib = _lib.load()
    N = 2
    if up and (H % 2 or W % 2):
        pytest.skip("upsampled sources need even sizes")
    xs = rnd(N, cin, H >> up, W >> up, seed=1, scale=1.5)
    sc, sh = rnd(cin, seed=2) * 0.3 + 1.0, rnd(cin, seed=3) * 0.4
    xin = F.relu(xs * sc[None, :, None, None] + sh[None, :, None, None])
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    g, y = rnd(N, cout, H, W, seed=4, scale=3e-4), rnd(N, cout, H, W, seed=5)
    a, b = rnd(cout, seed=6) * 0.2 + 1, rnd(cout, seed=7) * 0.2
    A, B, D = rnd(cout, seed=8) + 2.0, rnd(cout, seed=9) * 1e-5, rnd(cout, seed=10) * 1e-5
    if bnb:
        yh = y * a[None, :, None, None] + b[None, :, None, None]
        dy = torch.where(yh > 0, g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
        cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
        dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(y))
        amax = dev(torch.tensor([float((A.abs().max() * g.abs().max()))]))
    else:
        dy = g * a[None, :, None, None] + b[None, :, None, None] * 1e-4
        cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1] = a, b * 1e-4
        dsrc = make_src(dev(g), cout, SRC_AFFINE, act=ACT_NONE, cst=dev(cst))
        amax = dev(torch.tensor([float(dy.abs().max())]))
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), w, padding=1).backward(dy.double())
    args = sc_wgrad_args()
    args.dy = dsrc
    args.nsrc = 1
    args.src[0] = make_src(dev(xs), cin, SRC_AFFINE, act=ACT_RELU, up=up, cst=cst_affine(sc, sh))
    args.N, args.H, args.W, args.Cout, args.Cin, args.ks = N, H, W, cout, cin, 3
    n = lib.sc_wgrad_thin16_workspace_floats(N, H, W, cout, cin)
    ws = torch.empty(n, device=DEV); dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
    args.part, args.part_floats, args.dw = ws.data_ptr(), n, dw.data_ptr()
    args.terms, args.absmax = TERMS_F16X2, amax.data_ptr()
    check(lib.sc_conv3x3_wgrad_thin16(C.byref(args), stream()))
    assert relerr(dw, w.grad) < 2e-5


def test_wave_specialised_kernel_covers_every_feature():
    """k_conv3_ws normally takes only K loops of >= 16 chunks (decoder.blocks.0).  Re-run the split-kernel op tests in a child process
    with STARCOP_BX3_WS_MINCHUNKS=1, which routes EVERY two-fp16-term 3x3 launch through it: concat + upsampled sources, the
    BatchNorm-backward prologue, split outputs with the fused 2x2 down-sum, add tensors, statistics, odd chunk counts, ragged tiles."""
    import subprocess, sys
    env = dict(os.environ, STARCOP_BX3_WS_MINCHUNKS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "(bx3 or fp16 or two_term) and not wave_specialised"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_depthwise_plane_kernels_at_32():
    """The wave-per-plane depthwise kernels also exist for 32 x 32 planes (opt-in: measured slower there than the work-group form).
    Re-run the depthwise op tests -- they contain 32 x 32 stride-1 cases with N * C % 4 == 0 -- in a child process with
    STARCOP_DW_P32=1 so that the template stays correct for both sizes."""
    import subprocess, sys
    env = dict(os.environ, STARCOP_DW_P32="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "(dw_bwd_fused or test_depthwise) and not plane_kernels"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("npp", ["1", "2"])
def test_sub_pixel_tile_sizes_of_narrow_planes(npp):
    """16-wide sub-pixel tiles come in two heights: 16 rows (two pixel groups per wave) and, for launches with fewer than 32 such tiles
    (decoder.blocks.0 at batch 16), 8 rows (one group per wave, twice the work-groups).  The rule would pick the 8-row form for every
    narrow shape of this file's small-batch tests, so the sub-pixel op tests (forward, bf16 mode, data gradient incl. the one-launch
    skip gradient, fuzz) are re-run in child processes with either form forced (STARCOP_SP_NPP)."""
    import subprocess, sys
    env = dict(os.environ, STARCOP_SP_NPP=npp)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), os.path.join(here, "test_gpu_fuzz.py"), "-q", "-x", "-k",
                        "conv_sp and not wgrad and not tile_sizes"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ---- pointwise convolutions on the split-bf16 MFMA without LDS staging (conv_pw3.hip)
PW3_SHAPES = [(16, 96, 2, 64, 64), (64, 384, 3, 32, 32), (384, 64, 2, 32, 32), (160, 960, 2, 16, 16), (960, 320, 2, 16, 16), (24, 144, 2, 24, 40),
              (144, 24, 1, 20, 12), (320, 1280, 2, 4, 6), (32, 16, 1, 2, 2), (96, 576, 5, 2, 3), (40, 8, 2, 1, 1)]


@pytest.mark.parametrize("cin,cout,N,H,W", PW3_SHAPES)
def test_conv_pw3_forward_affine_stats(hip, cin, cout, N, H, W):
    """sc_conv1x1_pw3 forward: ReLU6(BatchNorm) prologue, fp32-accurate three-bf16-term contraction, per-32-pixel statistics rows;
    against F.conv2d in float64: <= 2e-6 of the largest output (one fp32 rounding per product + fp32 accumulation)"""
    from hip_ops import conv_pw3, pack_pw3
    x, w = rnd(N, cin, H, W, seed=1, scale=3.0), rnd(cout, cin, 1, 1, seed=2, scale=0.3)
    cst = torch.rand(cin, SC_CST, generator=torch.Generator().manual_seed(3)) + 0.5
    for mode, act in ((SRC_AFFINE, ACT_RELU6), (SRC_RAW, ACT_NONE)):
        src = make_src(dev(x), cin, mode, act=act, cst=dev(cst) if mode == SRC_AFFINE else None)
        xin = x.double() if mode == SRC_RAW else act_ref(x.double() * cst[:, 0].double()[None, :, None, None] + cst[:, 1].double()[None, :, None, None], act)
        ref = F.conv2d(xin, w.double())
        out, st = conv_pw3(src, pack_pw3(dev(w), 0), N, H, W, cout, want_stats=True)
        assert relerr(out, ref) < 2e-6, mode
        assert not torch.isnan(st).any()
        assert relerr(st.sum(0)[:, 0], ref.sum((0, 2, 3))) < 1e-5 and relerr(st.sum(0)[:, 1], (ref * ref).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("cin,cout,N,H,W", PW3_SHAPES)
def test_conv_pw3_dgrad_bnbwd_add_accum(hip, cin, cout, N, H, W):
    """backward-data: the BatchNorm / activation backward applied on load (g, y -> dy), transposed filter, the residual gradient
    added in the epilogue and accumulation into an existing gradient; against conv_transpose2d in float64"""
    from hip_ops import conv_pw3, pack_pw3
    w = rnd(cout, cin, 1, 1, seed=2, scale=0.3)
    g, y = rnd(N, cout, H, W, seed=3) * 1e-3, rnd(N, cout, H, W, seed=4)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6) * 0.3 + 1, rnd(cout, seed=7) * 1e-4, rnd(cout, seed=8) * 1e-4
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    gm = torch.where((yh > 0) & (yh < 6), g, torch.zeros(()))
    dy = gm.double() * A.double()[None, :, None, None] + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None]
    cstb = torch.zeros(cout, SC_CST); cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = a, b, A, B, D
    dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cstb), aux=dev(y))
    ref = F.conv_transpose2d(dy, w.double())
    wpk = pack_pw3(dev(w), 1)
    dx, _ = conv_pw3(dsrc, wpk, N, H, W, cin)
    assert relerr(dx, ref) < 1e-5            # dy itself is formed in fp32
    res = rnd(N, cin, H, W, seed=9) * 1e-3
    old = rnd(N, cin, H, W, seed=10) * 1e-3
    acc = dev(old.clone())
    dx2, _ = conv_pw3(dsrc, wpk, N, H, W, cin, add0=dev(res), accum_into=acc)
    assert relerr(dx2, ref + res.double() + old.double()) < 1e-5


@pytest.mark.parametrize("cin,cout,N,H,W", PW3_SHAPES)
def test_conv_pw3_dgrad_leaves_batchnorm_backward_sums(hip, cin, cout, N, H, W):
    """sc_bnr_args on sc_conv1x1_pw3 (round 6): a projection's data-gradient launch writes the COMPLETE gradient of the depthwise output,
    so its epilogue leaves that tensor's BatchNorm-backward sums {sum g', sum g' x_hat} (g' = dx relu6'(BN(y_in))), one row per 32-pixel
    block -- against float64 sums of the launch's own output, the gradient bit-identical to the launch without them, and through
    sc_bn_bwd_finalize_rows32 against sc_bn_bwd_small (what the network ran before) on the same (dx, y_in)"""
    import hip_ops
    from hip_ops import conv_pw3, pack_pw3
    w = rnd(cout, cin, 1, 1, seed=2, scale=0.3)
    g, y = rnd(N, cout, H, W, seed=3) * 1e-3, rnd(N, cout, H, W, seed=4)
    cstb = torch.zeros(cout, SC_CST)
    cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2, rnd(cout, seed=6) * 0.3 + 1, rnd(cout, seed=7) * 1e-4, rnd(cout, seed=8) * 1e-4
    dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cstb), aux=dev(y))
    wpk = pack_pw3(dev(w), 1)
    y_in = rnd(N, cin, H, W, seed=11) * 2
    cst_in = torch.zeros(cin, SC_CST)
    cst_in[:, 0], cst_in[:, 1], cst_in[:, 2], cst_in[:, 3] = rnd(cin, seed=12) * 0.2 + 1, rnd(cin, seed=13) * 0.5 + 1.5, rnd(cin, seed=14) * 0.1, rnd(cin, seed=15).abs() * 0.2 + 0.8
    dx0, _ = conv_pw3(dsrc, wpk, N, H, W, cin)
    dx1, _ = conv_pw3(dsrc, wpk, N, H, W, cin, bnr=(dev(y_in), dev(cst_in), ACT_RELU6))
    assert torch.equal(dx0, dx1)
    rows, _ = hip_ops.LAST_BNR
    assert bool(torch.isfinite(rows).all())
    sc, sh, mu, isd = (cst_in[:, k].double()[None, :, None, None] for k in range(4))
    yh = y_in.double() * sc + sh
    gp = torch.where((yh > 0) & (yh < 6), dx1.double().cpu(), torch.zeros((), dtype=torch.float64))
    s1, s2 = gp.sum((0, 2, 3)), (gp * (y_in.double() - mu) * isd).sum((0, 2, 3))
    got = rows.double().sum(0).cpu()
    scale = max(float(s1.abs().max()), float(s2.abs().max()), 1e-6)
    assert float((got[:, 0] - s1).abs().max()) < 2e-5 * scale and float((got[:, 1] - s2).abs().max()) < 2e-5 * scale
    out = {}
    for name in ("epilogue", "small"):
        dg, db, cb = (torch.full((n_,), float("nan"), device=DEV) for n_ in (cin, cin, cin * SC_CST))
        if name == "epilogue":
            check(hip.sc_bn_bwd_finalize_rows32(ptr(rows), rows.shape[0], float(N * H * W), ptr(dev(cst_in)), ptr(dg), ptr(db), ptr(cb), cin, None, stream()))
        else:
            check(hip.sc_bn_bwd_small(ptr(dx1), ptr(dev(y_in)), ptr(dev(cst_in)), ACT_RELU6, N, cin, H * W, ptr(dg), ptr(db), ptr(cb), None, None, stream()))
        out[name] = (dg.cpu(), db.cpu(), cb.cpu())
    for a_, b_ in zip(out["epilogue"], out["small"]):
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-6)


@pytest.mark.parametrize("cin,cout,N,H,W", [s for s in PW3_SHAPES if (s[3] * s[4]) % 8 == 0])
@pytest.mark.parametrize("deferred", [False, True])
def test_conv_pw3_wgrad(hip, cin, cout, N, H, W, deferred):
    """weight gradient with K = pixels: BNBWD dy and AFFINE input split in registers; against autograd in float64; both the
    in-place finish and the deferred batch reduction"""
    from hip_ops import wgrad_pw3
    x = rnd(N, cin, H, W, seed=1, scale=3.0)
    cst = torch.rand(cin, SC_CST, generator=torch.Generator().manual_seed(3)) + 0.5
    xin = act_ref(x.double() * cst[:, 0].double()[None, :, None, None] + cst[:, 1].double()[None, :, None, None], ACT_RELU6)
    g, y = rnd(N, cout, H, W, seed=3) * 1e-3, rnd(N, cout, H, W, seed=4)
    a, b = rnd(cout, seed=4) * 0.2 + 1, rnd(cout, seed=5) * 0.2
    A, B, D = rnd(cout, seed=6) * 0.3 + 1, rnd(cout, seed=7) * 1e-4, rnd(cout, seed=8) * 1e-4
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    gm = torch.where(yh > 0, g, torch.zeros(()))
    dy = gm.double() * A.double()[None, :, None, None] + B.double()[None, :, None, None] * y.double() + D.double()[None, :, None, None]
    cstb = torch.zeros(cout, SC_CST); cstb[:, 0], cstb[:, 1], cstb[:, 2], cstb[:, 3], cstb[:, 4] = a, b, A, B, D
    dsrc = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cstb), aux=dev(y))
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU6, cst=dev(cst))
    ref = torch.einsum("nohw,nihw->oi", dy, xin)[:, :, None, None]
    dw = wgrad_pw3(dsrc, src, N, H, W, cout, cin, deferred=deferred)
    assert relerr(dw, ref) < 1e-5




def test_cast_f64_f32_batch(hip):
    """sc_cast_f64_f32_batch: every (fp64 accumulator -> fp32 gradient view) pair of a backward walk in one launch == the per-layer casts"""
    import numpy as np
    g = torch.Generator().manual_seed(3)
    acc = (torch.randn(5000, generator=g, dtype=torch.float64) * 1e3).to(DEV)
    out = torch.full((6000,), float("nan"), device=DEV)
    # (offset in acc, offset in out, n): ragged sizes, one longer than 8 x 256 items, destinations not in source order
    spans = [(0, 4000, 288), (288, 0, 2700), (2988, 3000, 9), (2997, 5000, 864)]
    descs = np.array([(acc.data_ptr() + 8 * a, out.data_ptr() + 4 * b, n) for a, b, n in spans], dtype=np.uint64)
    tab = torch.from_numpy(descs.view(np.uint8).copy()).to(DEV)
    check(hip.sc_cast_f64_f32_batch(ptr(tab), len(spans), stream()))
    want = torch.full((6000,), float("nan"), device=DEV)
    for a, b, n in spans:
        want[b:b + n] = acc[a:a + n].float()
    assert torch.equal(torch.nan_to_num(out, nan=-7.0), torch.nan_to_num(want, nan=-7.0))      # nothing else was written


def test_stream_wait_stream_orders_two_streams(hip):
    """sc_stream_wait_stream (events without the system-scope fence): work queued on the waiter after the call sees everything the
    signaller had queued before it -- a long producer on stream A, a consumer on stream B, 50 rounds, both directions"""
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 24, device=DEV)
    y = torch.zeros(1 << 24, device=DEV)
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    ha, hb = C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream)
    for k in range(50):
        with torch.cuda.stream(a):
            x.add_(1.0)                                     # producer (64 MB read + write)
        check(hip.sc_stream_wait_stream(hb, ha))            # b waits for a
        with torch.cuda.stream(b):
            y.copy_(x)                                      # consumer must see round k + 1 everywhere
            bad += (y != float(k + 1)).sum()
        check(hip.sc_stream_wait_stream(ha, hb))            # a's next add must not overtake the copy
    torch.cuda.synchronize()
    assert int(bad) == 0 and float(y.min()) == 50.0 and float(y.max()) == 50.0
