"""Seeded random shapes through the MFMA convolution entry points (fp32-MFMA, split-bf16, split-K 1x1; forward, dgrad and
wgrad) against torch fp64 references: ragged H/W (not multiples of the 8x32 / 4x32 tiles), channel counts that are not
multiples of the K chunk or the cout tile, batch 1..3, two-source concat with an upsampled first source."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_ops import dev, conv_mfma, cst_affine, pack, pack_bx3, relerr, wgrad_mfma  # noqa: E402
from starcop_amd._lib import ACT_RELU, ACT_RELU6, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_RAW, make_src  # noqa: E402


def _rnd(rng, *shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * scale)


@pytest.fixture(params=[3, 4], ids=["bf16x3", "fp16x2"])
def split_mode(request):
    """run a split-kernel test under both operand splits (three bf16 terms / two fp16 terms)"""
    import hip_ops
    old, hip_ops.DEFAULT_BX3_TERMS = hip_ops.DEFAULT_BX3_TERMS, request.param
    yield request.param
    hip_ops.DEFAULT_BX3_TERMS = old



@pytest.mark.parametrize("seed", range(12))
def test_fuzz_conv3(hip, split_mode, seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(3, 45)), int(rng.integers(3, 75))
    cin = int(rng.choice([8, 16, 24, 40, 56, 72, 104]))
    cout = int(rng.choice([16, 24, 32, 40, 64, 72, 136]))
    x, w = _rnd(rng, N, cin, H, W), _rnd(rng, cout, cin, 3, 3, scale=0.2)
    sc, sh = _rnd(rng, cin) * 0.3 + 1.0, _rnd(rng, cin) * 0.2
    xin = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    src = make_src(dev(x), cin, SRC_AFFINE, act=ACT_RELU, cst=cst_affine(sc, sh))
    co_t = 16 if cout <= 16 else (32 if cout <= 32 else 64)
    (o32,), st32 = conv_mfma([src], pack(dev(w), co_t, 0), N, H, W, cout, 3, co_t, want_stats=True)
    assert relerr(o32, ref) < 1e-5
    assert relerr(st32.double().sum(0)[:, 0], ref.sum((0, 2, 3))) < 1e-4
    if cout > 16:
        cx = 32 if cout <= 32 else 64
        (ox,), stx = conv_mfma([src], pack_bx3(dev(w), cx, 0), N, H, W, cout, 3, cx, want_stats=True, bx3=True)
        assert relerr(ox, ref) < 1e-5
        assert relerr(stx.double().sum(0)[:, 1], (ref ** 2).sum((0, 2, 3))) < 1e-4
    # backward-data + weight gradient from a BatchNorm-backward source
    g, y = _rnd(rng, N, cout, H, W), _rnd(rng, N, cout, H, W)
    a, b = _rnd(rng, cout) * 0.2 + 1, _rnd(rng, cout) * 0.2
    A, B, D = _rnd(rng, cout), _rnd(rng, cout) * 0.1, _rnd(rng, cout) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where((yh > 0) & (yh < 6), g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    dys = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU6, cst=dev(cst), aux=dev(y))
    refb = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    cb = 16 if cin <= 16 else (32 if cin <= 32 else 64)
    (d32,), _ = conv_mfma([dys], pack(dev(w), cb, 1), N, H, W, cin, 3, cb)
    assert relerr(d32, refb) < 2e-5
    if cin > 16:
        cbx = 32 if cin <= 32 else 64
        (dx,), _ = conv_mfma([dys], pack_bx3(dev(w), cbx, 1), N, H, W, cin, 3, cbx, bx3=True)
        assert relerr(dx, refb) < 2e-5
    wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), wz, padding=1).backward(dy.double())
    assert relerr(wgrad_mfma(dys, [src], N, H, W, cout, cin, 3), wz.grad) < 2e-5
    if cout >= 32 and cin >= 32:
        assert relerr(wgrad_mfma(dys, [src], N, H, W, cout, cin, 3, bx3=True), wz.grad) < 2e-5


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_conv1(hip, seed):
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 20)), int(rng.integers(1, 30))
    cin = int(rng.choice([8, 24, 96, 200, 392]))
    cout = int(rng.choice([8, 24, 40, 96, 168]))
    x, w = _rnd(rng, N, cin, H, W), _rnd(rng, cout, cin, 1, 1, scale=0.2)
    ref = F.conv2d(x.double(), w.double())
    co_t = 32 if cout <= 32 else 64
    wp = pack(dev(w), co_t, 0)
    src = make_src(dev(x), cin, SRC_RAW)
    for ks in (False, True):
        (o,), st = conv_mfma([src], wp, N, H, W, cout, 1, co_t, want_stats=True, ksplit=ks)
        assert relerr(o, ref) < 1e-5
        assert relerr(st.double().sum(0)[:, 0], ref.sum((0, 2, 3))) < 1e-4
    wz = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, requires_grad=True)
    g = _rnd(rng, N, cout, H, W)
    F.conv2d(x.double(), wz).backward(g.double())
    assert relerr(wgrad_mfma(make_src(dev(g), cout, SRC_RAW), [src], N, H, W, cout, cin, 1), wz.grad) < 2e-5


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_concat_upsample(hip, split_mode, seed):
    rng = np.random.default_rng(3000 + seed)
    N = int(rng.integers(1, 3))
    H, W = 2 * int(rng.integers(2, 14)), 2 * int(rng.integers(2, 30))
    c0, c1 = 16 * int(rng.integers(1, 5)), 8 * int(rng.integers(1, 6))
    cout = int(rng.choice([32, 48, 64, 80]))
    prev, skip = _rnd(rng, N, c0, H // 2, W // 2), _rnd(rng, N, c1, H, W)
    w = _rnd(rng, cout, c0 + c1, 3, 3, scale=0.1)
    sc0, sh0 = _rnd(rng, c0) * 0.3 + 1, _rnd(rng, c0) * 0.2
    xin = torch.cat([F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2, mode="nearest"), skip], 1)
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    s0 = make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))
    s1 = make_src(dev(skip), c1, SRC_RAW)
    co_t = 32 if cout <= 32 else 64
    (o,), _ = conv_mfma([s0, s1], pack_bx3(dev(w), co_t, 0), N, H, W, cout, 3, co_t, bx3=True)
    assert relerr(o, ref) < 1e-5
    g = _rnd(rng, N, cout, H, W)
    wz = torch.zeros(cout, c0 + c1, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), wz, padding=1).backward(g.double())
    dys = make_src(dev(g), cout, SRC_RAW)
    assert relerr(wgrad_mfma(dys, [s0, s1], N, H, W, cout, c0 + c1, 3, bx3=True), wz.grad) < 2e-5
    assert relerr(wgrad_mfma(dys, [s0, s1], N, H, W, cout, c0 + c1, 3), wz.grad) < 2e-5


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_conv_sp(hip, seed):
    """seeded random decoder-conv1 shapes through the sub-pixel kernels (sc_conv3x3_sp, sc_conv3x3_sp_dgrad) against float64 of the
    reference's op sequence (interpolate(nearest, x2) -> cat -> conv3x3 and its autograd): ragged low-resolution planes of both tile
    shapes, channel counts off the 16-channel chunk / 32-channel block / 128-channel tile, with and without skip channels"""
    from hip_ops import conv_sp, conv_sp_dgrad, pack_sp, pack_spd
    from starcop_amd import _lib
    from starcop_amd._lib import ACT_NONE
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.integers(1, 4))
    Hl, Wl = int(rng.integers(2, 24)), int(rng.integers(2, 45))
    H, W = 2 * Hl, 2 * Wl
    c0 = int(rng.choice([8, 16, 24, 40, 64, 72, 136]))
    c1 = int(rng.choice([0, 8, 12, 16, 24, 40]))
    cout = int(rng.choice([16, 24, 32, 40, 64, 72]))
    prev, w = _rnd(rng, N, c0, Hl, Wl), _rnd(rng, cout, c0 + c1, 3, 3, scale=0.2)
    sc0, sh0 = _rnd(rng, c0) * 0.3 + 1.0, _rnd(rng, c0) * 0.2
    up = F.interpolate(F.relu(prev * sc0[None, :, None, None] + sh0[None, :, None, None]), scale_factor=2, mode="nearest")
    srcs = [make_src(dev(prev), c0, SRC_AFFINE, act=ACT_RELU, up=1, cst=cst_affine(sc0, sh0))]
    xin = up
    if c1:
        skip = _rnd(rng, N, c1, H, W)
        sc1, sh1 = _rnd(rng, c1) * 0.3 + 1.0, _rnd(rng, c1) * 0.2
        xin = torch.cat([up, skip * sc1[None, :, None, None] + sh1[None, :, None, None]], 1)
        srcs.append(make_src(dev(skip), c1, SRC_AFFINE, act=ACT_NONE, cst=cst_affine(sc1, sh1)))
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    wd = dev(w)
    out, stats = conv_sp(srcs, pack_sp(wd, c0, batched=bool(seed & 1)), N, H, W, cout, want_stats=True)
    assert relerr(out, ref) < 1e-5
    st = stats.double().sum(0).cpu()
    assert relerr(st[:, 0], ref.sum((0, 2, 3))) < 1e-4 and relerr(st[:, 1], (ref ** 2).sum((0, 2, 3))) < 1e-4
    # data gradient of the up-sampled channels (and, where the tile has room, of the skip channels) from a BatchNorm-backward source
    g, y = _rnd(rng, N, cout, H, W), _rnd(rng, N, cout, H, W)
    a, b = _rnd(rng, cout) * 0.2 + 1, _rnd(rng, cout) * 0.2
    A, B, D = _rnd(rng, cout), _rnd(rng, cout) * 0.1, _rnd(rng, cout) * 0.1
    yh = y * a[None, :, None, None] + b[None, :, None, None]
    dy = torch.where(yh > 0, g, torch.zeros(())) * A[None, :, None, None] + B[None, :, None, None] * y + D[None, :, None, None]
    cst = torch.zeros(cout, SC_CST); cst[:, 0], cst[:, 1], cst[:, 2], cst[:, 3], cst[:, 4] = a, b, A, B, D
    dys = make_src(dev(g), cout, SRC_BNBWD, act=ACT_RELU, cst=dev(cst), aux=dev(y))
    full = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    ref_up = F.avg_pool2d(full[:, :c0], 2) * 4
    amax = torch.tensor([float((A[None, :, None, None] * g).abs().max())], device="cuda")
    vs = bool(c1) and bool(_lib.load().sc_spd_vskip_ok(c0, c1))
    st_ = bool(c1) and not vs and bool(seed & 2)          # skip tiles: the skip channels as additional channel tiles of the launch
    wpk = pack_spd(wd, c0, batched=not (seed & 1), vskip=vs, skip_tiles=st_)
    if vs or st_:
        d_up, d_sk = conv_sp_dgrad(dys, wpk, N, H, W, c0, absmax=amax, cskip=c1)
        assert relerr(d_sk, full[:, c0:]) < 1e-5
    else:
        d_up = conv_sp_dgrad(dys, wpk, N, H, W, c0, absmax=amax)
    assert relerr(d_up, ref_up) < 1e-5
