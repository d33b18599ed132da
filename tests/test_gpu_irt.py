"""Fused TRAINING execution of an inverted-residual block's expansion + depthwise pair (csrc/conv_irt.hip: the 6x-expanded tensor
is never stored) against a float64 evaluation of the torch ops the reference dispatches for it -- F.conv2d 1x1, train-mode
F.batch_norm, relu6, depthwise F.conv2d and their autograd (torchvision InvertedResidual inside smp.Unet('mobilenet_v2'),
/root/reference/starcop/models/model_module.py:244-251).  Every sweep goes through the C ABI."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_ops import DEV, cst_affine, dev, relerr  # noqa: E402
from starcop_amd import _lib  # noqa: E402
from starcop_amd._lib import ACT_NONE, ACT_RELU6, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_RAW, check, make_src, ptr, sc_irt_args, stream  # noqa: E402

EPS = 1e-5

# (N, Cin, hidden, H, W, stride, input source): the shapes of features.2 .. features.7 scaled down, plus ragged tiles
# (H not a multiple of 8, W not a multiple of 32), hidden not a multiple of 32 and every Cin the kernels take
CASES = [
    (2, 16, 96, 32, 64, 2, "affine"),
    (2, 24, 144, 16, 64, 1, "affine"),
    (1, 24, 144, 24, 40, 2, "raw"),
    (2, 32, 192, 20, 40, 1, "raw"),
    (1, 8, 40, 12, 24, 1, "affine"),
    (1, 16, 80, 36, 72, 2, "raw"),
]


def _reference(x, xs, xh, We, Wd, gam, bet, R, stride):
    """float64 torch: forward tensors and autograd gradients of loss = sum(d * R)"""
    xa = (x.double() * xs.double()[None, :, None, None] + xh.double()[None, :, None, None]).requires_grad_(True)
    We64, Wd64 = We.double().requires_grad_(True), Wd.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    e = F.conv2d(xa, We64[:, :, None, None])
    eh = F.relu6(F.batch_norm(e, None, None, g64, b64, training=True, eps=EPS))
    d = F.conv2d(eh, Wd64[:, None], stride=stride, padding=1, groups=We.shape[0])
    (d * R.double()).sum().backward()
    return dict(e=e.detach(), d=d.detach(), dx=xa.grad, dWe=We64.grad, dWd=Wd64.grad, dgam=g64.grad, dbet=b64.grad)


def _switch_margin(x, xs, xh, We, gam, bet):
    xa = x.double() * xs.double()[None, :, None, None] + xh.double()[None, :, None, None]
    y = F.batch_norm(F.conv2d(xa, We.double()[:, :, None, None]), None, None, gam.double(), bet.double(), training=True, eps=EPS)
    return float(torch.minimum(y.abs(), (y - 6).abs()).min())


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}_Cin{c[1]}_Hd{c[2]}_{c[3]}x{c[4]}_s{c[5]}_{c[6]}" for c in CASES])
def test_irt_sweeps_vs_fp64(hip, case):
    lib = hip
    N, Cin, Hd, H, W, S, srcmode = case
    assert lib.sc_irt_supported(Cin, Hd, H, W, S) == 1
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    for seed in range(100 + Cin + Hd + H, 100 + Cin + Hd + H + 200):
        # a ReLU6 switch within rounding of flipping is a DISCRETE difference between an fp32 and an fp64 evaluation (one flipped
        # element moves a dx entry by O(1)): advance the seed until no BatchNorm output lies within 2e-5 of 0 or 6
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2
        if srcmode == "affine":
            xs, xh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        else:
            xs, xh = torch.ones(Cin), torch.zeros(Cin)
        We = torch.randn(Hd, Cin, generator=g) * (2.0 / Cin) ** 0.5
        Wd = torch.randn(Hd, 3, 3, generator=g) * 0.4
        gam, bet = torch.rand(Hd, generator=g) + 0.5, torch.randn(Hd, generator=g) * 0.5 + 1.0
        R = torch.randn(N, Hd, Ho, Wo, generator=g)
        if _switch_margin(x, xs, xh, We, gam, bet) > 2e-5:
            break
    else:
        raise AssertionError("no seed with a safe ReLU6 margin")
    ref = _reference(x, xs, xh, We, Wd, gam, bet, R, S)

    xd, Wed, Wdd = dev(x), dev(We), dev(Wd)
    a = sc_irt_args()
    a.x = make_src(xd, Cin, SRC_AFFINE, act=ACT_NONE, cst=cst_affine(xs, xh)) if srcmode == "affine" else make_src(xd, Cin, SRC_RAW)
    a.w_expand, a.w_dw = Wed.data_ptr(), Wdd.data_ptr()
    cst_e = torch.zeros(Hd, SC_CST, device=DEV)
    a.cst_expand = cst_e.data_ptr()
    a.N, a.Cin, a.hidden, a.H, a.W, a.stride = N, Cin, Hd, H, W, S
    st = stream()
    # ---- (A) statistics of e -> train-mode BatchNorm constants
    rows0 = lib.sc_irt_rows(0, N, H, W, S)
    stats = torch.full((rows0, Hd, 2), float("nan"), device=DEV)
    check(lib.sc_irt_expand_stats(C.byref(a), ptr(stats), st))
    tot = stats.double().sum(0).cpu()
    e_ref = ref["e"]
    assert relerr(tot[:, 0], e_ref.sum((0, 2, 3))) < 2e-5 and relerr(tot[:, 1], (e_ref ** 2).sum((0, 2, 3))) < 2e-5
    gd, bd = dev(gam), dev(bet)
    rm, rv = torch.zeros(Hd, device=DEV), torch.ones(Hd, device=DEV)
    check(lib.sc_bn_finalize(ptr(stats), rows0, float(N * H * W), ptr(gd), ptr(bd), ptr(rm), ptr(rv), 0.1, EPS, 1, ptr(cst_e), Hd, None, st))
    mean_ref, var_ref = e_ref.mean((0, 2, 3)), e_ref.var((0, 2, 3), unbiased=False)
    assert relerr(cst_e[:, 2], mean_ref) < 1e-5 and relerr(cst_e[:, 3], 1.0 / torch.sqrt(var_ref + EPS)) < 1e-5
    # ---- (B) forward: raw depthwise output + its statistics rows
    rows1 = lib.sc_irt_rows(1, N, H, W, S)
    d_out = torch.full((N, Hd, Ho, Wo), float("nan"), device=DEV)
    stats_d = torch.full((rows1, Hd, 2), float("nan"), device=DEV)
    check(lib.sc_irt_fwd(C.byref(a), ptr(d_out), ptr(stats_d), st))
    e_d = relerr(d_out, ref["d"])
    totd = stats_d.double().sum(0).cpu()
    assert e_d < 1e-5, e_d
    assert relerr(totd[:, 0], ref["d"].sum((0, 2, 3))) < 2e-5 and relerr(totd[:, 1], (ref["d"] ** 2).sum((0, 2, 3))) < 2e-5
    # ---- (Bi) backward sums
    Rd = dev(R)
    dy = make_src(Rd, Hd, SRC_RAW)
    rows2 = lib.sc_irt_bwd_rows(N, Hd, H, W)
    esums = torch.full((rows2, Hd, 2), float("nan"), dtype=torch.float64, device=DEV)
    dwacc = torch.zeros(Hd, 9, dtype=torch.float64, device=DEV)
    work = torch.zeros(lib.sc_irt_bwd_workspace_floats(N, Hd, H, W), device=DEV)
    check(lib.sc_irt_bwd_sums(C.byref(a), C.byref(dy), ptr(esums), ptr(dwacc), ptr(work), st))
    e_wd = relerr(dwacc.reshape(Hd, 3, 3), ref["dWd"])
    cstb = torch.zeros(Hd, SC_CST, device=DEV)
    dgam, dbet = torch.empty(Hd, device=DEV), torch.empty(Hd, device=DEV)
    check(lib.sc_bn_bwd_finalize(ptr(esums), rows2, float(N * H * W), ptr(cst_e), ptr(dgam), ptr(dbet), ptr(cstb), Hd, st))
    e_g, e_b = relerr(dgam, ref["dgam"]), relerr(dbet, ref["dbet"])
    # ---- (Bii) backward data
    dx = torch.full((N, Cin, H, W), float("nan"), device=DEV)
    check(lib.sc_irt_bwd_data(C.byref(a), C.byref(dy), ptr(cstb), ptr(dx), None, 0, st))
    e_x = relerr(dx, ref["dx"])
    # residual add + accumulate variants of the store
    add = torch.randn(N, Cin, H, W, device=DEV)
    dx2 = dx.clone()
    check(lib.sc_irt_bwd_data(C.byref(a), C.byref(dy), ptr(cstb), ptr(dx2), ptr(add), 1, st))
    e_x2 = relerr(dx2, 2 * ref["dx"] + add.cpu().double())
    # ---- expansion filter gradient from the partial rows
    dWe = torch.full((Hd, Cin), float("nan"), device=DEV)
    check(lib.sc_irt_wgrad_finalize(C.byref(a), ptr(cstb), ptr(work), ptr(dWe), st))
    e_we = relerr(dWe, ref["dWe"])
    print(f"irt {case}: d {e_d:.1e}  dW_dw {e_wd:.1e}  dgamma {e_g:.1e}  dbeta {e_b:.1e}  dx {e_x:.1e} / {e_x2:.1e}  dW_e {e_we:.1e}")
    assert e_wd < 2e-5 and e_g < 2e-5 and e_b < 2e-5 and e_x < 2e-5 and e_x2 < 2e-5 and e_we < 2e-5


def test_irt_bnbwd_gradient_source(hip):
    """the gradient of the depthwise output arrives as an SC_SRC_BNBWD source in the network (g w.r.t. the activated output, the raw
    d and the constants of BN_d's backward): both backward sweeps must apply dy = A [pass] g + B d + D on load"""
    lib = hip
    N, Cin, Hd, H, W, S = 2, 16, 96, 16, 32, 1
    for seed in range(5, 205):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(N, Cin, H, W, generator=g)
        We = torch.randn(Hd, Cin, generator=g) * 0.3
        Wd = torch.randn(Hd, 3, 3, generator=g) * 0.4
        gam, bet = torch.rand(Hd, generator=g) + 0.5, torch.randn(Hd, generator=g) * 0.5 + 1.0
        if _switch_margin(x, torch.ones(Cin), torch.zeros(Cin), We, gam, bet) > 2e-5:
            break
    gd_ = torch.randn(N, Hd, H, W, generator=g)                       # gradient w.r.t. relu6(BN_d(d))
    draw = torch.randn(N, Hd, H, W, generator=g) * 2 + 1              # "raw d" as the aux tensor
    cb = torch.zeros(Hd, SC_CST)
    cb[:, 0], cb[:, 1] = torch.rand(Hd, generator=g) + 0.5, torch.randn(Hd, generator=g)
    cb[:, 2], cb[:, 3], cb[:, 4] = torch.randn(Hd, generator=g), torch.randn(Hd, generator=g) * 0.1, torch.randn(Hd, generator=g) * 0.1
    yh = draw * cb[:, 0][None, :, None, None] + cb[:, 1][None, :, None, None]
    R = torch.where((yh > 0) & (yh < 6), gd_, torch.zeros(())) * cb[:, 2][None, :, None, None] + draw * cb[:, 3][None, :, None, None] \
        + cb[:, 4][None, :, None, None]
    ref = _reference(x, torch.ones(Cin), torch.zeros(Cin), We, Wd, gam, bet, R, S)
    xd, Wed, Wdd = dev(x), dev(We), dev(Wd)
    a = sc_irt_args()
    a.x = make_src(xd, Cin, SRC_RAW)
    a.w_expand, a.w_dw = Wed.data_ptr(), Wdd.data_ptr()
    e = ref["e"]
    mean, var = e.mean((0, 2, 3)), e.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + EPS)
    cst_e = torch.zeros(Hd, SC_CST, dtype=torch.float64)
    cst_e[:, 0], cst_e[:, 1], cst_e[:, 2], cst_e[:, 3] = gam.double() * invstd, bet.double() - mean * gam.double() * invstd, mean, invstd
    cst_e = dev(cst_e.float())
    a.cst_expand = cst_e.data_ptr()
    a.N, a.Cin, a.hidden, a.H, a.W, a.stride = N, Cin, Hd, H, W, S
    dy = make_src(dev(gd_), Hd, SRC_BNBWD, act=ACT_RELU6, cst=dev(cb), aux=dev(draw))
    st = stream()
    rows2 = lib.sc_irt_bwd_rows(N, Hd, H, W)
    esums = torch.zeros(rows2, Hd, 2, dtype=torch.float64, device=DEV)
    dwacc = torch.zeros(Hd, 9, dtype=torch.float64, device=DEV)
    work = torch.zeros(lib.sc_irt_bwd_workspace_floats(N, Hd, H, W), device=DEV)
    check(lib.sc_irt_bwd_sums(C.byref(a), C.byref(dy), ptr(esums), ptr(dwacc), ptr(work), st))
    cstb = torch.zeros(Hd, SC_CST, device=DEV)
    dgam, dbet = torch.empty(Hd, device=DEV), torch.empty(Hd, device=DEV)
    check(lib.sc_bn_bwd_finalize(ptr(esums), rows2, float(N * H * W), ptr(cst_e), ptr(dgam), ptr(dbet), ptr(cstb), Hd, st))
    dx = torch.empty(N, Cin, H, W, device=DEV)
    check(lib.sc_irt_bwd_data(C.byref(a), C.byref(dy), ptr(cstb), ptr(dx), None, 0, st))
    dWe = torch.empty(Hd, Cin, device=DEV)
    check(lib.sc_irt_wgrad_finalize(C.byref(a), ptr(cstb), ptr(work), ptr(dWe), st))
    errs = [relerr(dwacc.reshape(Hd, 3, 3), ref["dWd"]), relerr(dgam, ref["dgam"]), relerr(dbet, ref["dbet"]), relerr(dx, ref["dx"]),
            relerr(dWe, ref["dWe"])]
    print("irt BNBWD source: dW_dw dgamma dbeta dx dW_e", " ".join(f"{v:.1e}" for v in errs))
    assert max(errs) < 2e-5
