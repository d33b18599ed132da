"""G9: the HIP convolution kernels (forward, backward-data, backward-weight, bias/ReLU prologues), called through the C ABI,
against numbers the REFERENCE ITSELF produced with its in-repo conv blocks -- layer_factory.double_conv and UNet sub-blocks
(/root/reference/starcop/models/architectures/layer_factory.py:4-9, unet.py:7-51; fixture tests/golden/g9_convblocks.npz
written by tests/golden/make_golden.py::g9_convblocks).  This is the pin of every conv family to reference-executed
arithmetic: fp32-MFMA kernels (sc_conv2d_mfma / sc_conv2d_wgrad_mfma), and the split 16-bit-MFMA kernels
(sc_conv3x3_bx3 / sc_conv3x3_wgrad_bx3) under both operand splits.

A ``conv(bias) -> ReLU`` of the reference maps onto the product's "normalise on load" model as: the conv stores its raw output
y, consumers read relu(1*y + bias) through an SC_SRC_AFFINE prologue, and the backward reads dL/dy through SC_SRC_BNBWD with
(A, B, D) = (1, 0, 0); the bias gradient is the first column of sc_bn_bwd_reduce's sums."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import g9_util  # noqa: E402
from hip_ops import DEV, conv_mfma, dev, pack, pack_bx3, wgrad_mfma  # noqa: E402
from starcop_amd import _lib  # noqa: E402
from starcop_amd._lib import (ACT_NONE, ACT_RELU, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_RAW, STAT_BNBWD, check, make_src, ptr,
                              stream)  # noqa: E402

TOL = 1e-4      # north_star: within 1e-4 relative of the reference CPU path


def _cot(cout, ks):
    if cout <= 16 and ks == 3:
        return 16
    return 32 if cout <= 32 else 64


def _conv(srcs, w, N, H, W, tflip, terms, absmax=None):
    """forward (tflip=0) or backward-data (tflip=1: output channels = w.shape[1]) of `w` [co][ci][k][k] on the chosen kernel family"""
    ks = w.shape[2]
    cout = w.shape[1] if tflip else w.shape[0]
    if terms and ks == 3 and cout >= 32:
        co_t = _cot(cout, ks)
        (out,), _ = conv_mfma(srcs, pack_bx3(w, co_t, tflip, terms), N, H, W, cout, 3, co_t, bx3=True, terms=terms, absmax=absmax)
        return out
    co_t = _cot(cout, ks)
    (out,), _ = conv_mfma(srcs, pack(w, co_t, tflip), N, H, W, cout, ks, co_t)
    return out


def _bias_relu_cst(b, act):
    c = torch.zeros(b.numel(), SC_CST, device=DEV)
    c[:, 0], c[:, 1] = 1.0, b.to(DEV)          # forward prologue: act(1*y + b)
    c[:, 2], c[:, 3], c[:, 4] = 1.0, 0.0, 0.0  # backward prologue: dy = 1*[pass ? g : 0] + 0*y + 0
    return dev(c)


def _bias_grad_and_absmax(g, y, cst, act, N, Cc, HW):
    """sum over pixels of g*act'(y+b) per channel (= bias gradient) and the range hint of the two-fp16-term kernels"""
    lib = _lib.load()
    rows = lib.sc_stat_rows(STAT_BNBWD, N, int(HW ** 0.5), int(HW ** 0.5))
    sums = dev(torch.zeros(rows, Cc, 2, dtype=torch.float64))
    # xhat = (y - c[2]) * c[3] is only used for the second column; cst_fwd layout {scale, shift, mean, invstd}
    cf = cst.clone(); cf[:, 2], cf[:, 3] = 0.0, 1.0
    amax = dev(torch.zeros(1))
    check(lib.sc_bn_bwd_reduce(ptr(g), ptr(y), ptr(dev(cf)), act, ptr(sums), N, Cc, HW, ptr(amax), None, stream()))
    return sums.sum(0)[:, 0].float(), amax


_IDS = {0: "fp32mfma", 3: "split-bf16x3", 4: "split-fp16x2"}
# the split 16-bit-MFMA kernels take 3x3 layers with >= 32 channels: dc_4_8 (8 channels) and conv_last (1x1) have none
_COMBOS = [(n, t) for n in ("dc_4_8", "unet_down1", "unet_down2", "unet_up1", "unet_last") for t in (0, 3, 4)
           if t == 0 or n not in ("dc_4_8", "unet_last")]


@pytest.mark.parametrize("name,terms", _COMBOS, ids=[f"{n}-{_IDS[t]}" for n, t in _COMBOS])
def test_g9_blocks_hip(hip, name, terms):
    z = g9_util.load()
    params, x, r = g9_util.case_tensors(name)
    convs, xshape = g9_util.CASES[name]
    N, _, H, W = xshape
    last_act = ACT_NONE if name == "unet_last" else ACT_RELU
    cin0 = xshape[1]
    if cin0 % 8:        # the dense kernels read sources in multiples of 8 channels (the product's 4-channel input goes through the
        pad = 8 - cin0 % 8   # stem kernel): zero channels with zero filters add exact zeros to every sum
        x = torch.cat([x, torch.zeros(N, pad, H, W)], 1)
        w0, b0 = params[0]
        params[0] = (torch.cat([w0, torch.zeros(w0.shape[0], pad, *w0.shape[2:])], 1), b0)
        xshape = tuple(x.shape)
    cout_last = params[-1][0].shape[0]
    if cout_last % 8:   # likewise the backward-data kernel reads dL/dy in multiples of 8 channels: zero filters / bias / gradient
        pad = 8 - cout_last % 8
        wl, bl = params[-1]
        params[-1] = (torch.cat([wl, torch.zeros(pad, *wl.shape[1:])], 0), torch.cat([bl, torch.zeros(pad)]))
        r = torch.cat([r, torch.zeros(N, pad, H, W)], 1)
    xd, rd = dev(x), dev(r)
    if name == "unet_up1":      # the reference's cat([x_up, skip]) input: read as two concatenated sources, like smp's decoder conv1
        srcs = [make_src(dev(x[:, :128]), 128, SRC_RAW), make_src(dev(x[:, 128:]), 64, SRC_RAW)]
    else:
        srcs = [make_src(xd, xshape[1], SRC_RAW)]
    # ---- forward: raw conv outputs ys[i]; bias + ReLU live in the consumer prologue
    ys, csts, ins = [], [], []
    cur = srcs
    for i, (w, b) in enumerate(params):
        act = ACT_RELU if i + 1 < len(params) else last_act
        ins.append(cur)
        y = _conv(cur, dev(w), N, H, W, 0, terms)
        cst = _bias_relu_cst(b, act)
        ys.append(y); csts.append((cst, act))
        cur = [make_src(y, w.shape[0], SRC_AFFINE, act=act, cst=cst)]
    lib = _lib.load()
    out = torch.empty(N, params[-1][0].shape[0], H, W, device=DEV)
    check(lib.sc_apply_src(C.byref(cur[0]), ptr(out), N, out.shape[1], H * W, stream()))
    errs = {"y": g9_util.golden_err(z, name, "y", out[:, :cout_last])}
    # ---- backward
    g = rd
    for i in range(len(params) - 1, -1, -1):
        w, b = params[i]
        cst, act = csts[i]
        co, ci = w.shape[0], w.shape[1]
        gb, amax = _bias_grad_and_absmax(g, ys[i], cst, act, N, co, H * W)
        last = i == len(params) - 1
        errs[f"gb{i}"] = g9_util.golden_err(z, name, f"gb{i}", gb[:cout_last] if last else gb)
        dy = make_src(g, co, SRC_BNBWD, act=act, cst=cst, aux=ys[i])
        use_bx3 = bool(terms) and w.shape[2] == 3 and co >= 32 and ci >= 32
        gw = wgrad_mfma(dy, ins[i], N, H, W, co, ci, w.shape[2], bx3=use_bx3, terms=terms if use_bx3 else 0,
                        absmax=amax if terms == 4 else None)
        errs[f"gw{i}"] = g9_util.golden_err(z, name, f"gw{i}", (gw[:cout_last] if last else gw)[:, :cin0] if i == 0 else (gw[:cout_last] if last else gw))
        g = _conv([dy], dev(w), N, H, W, 1, terms, absmax=amax if terms == 4 else None)
    errs["gx"] = g9_util.golden_err(z, name, "gx", g[:, :cin0])
    print(f"G9 {name} terms={terms}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g9_full_unet_hip(hip, tag):
    """the reference's whole in-repo UNet(4, 1) on the HIP kernels (starcop_amd/unet_simple.py: split-MFMA 3x3
    convolutions, bias + ReLU prologues, sc_maxpool2x2, sc_upsample_bilinear2x, two-source concat convolutions, 1x1 head) against
    the logits the REFERENCE computed for the same weights and input -- a 15-convolution-deep pin of the kernels"""
    from starcop_amd.unet_simple import SimpleUNet
    net = SimpleUNet(4, 1)
    sd = g9_util.full_unet_state()
    assert list(net.state_dict().keys()) == list(sd.keys())            # the reference module's key names and order
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = g9_util.full_unet_input(tag)
    got = net(x.to(DEV))
    want = torch.from_numpy(g9_util.load()[f"unet_full.{tag}.y"])
    e = float((got.cpu() - want).abs().max() / want.abs().max())
    print(f"in-repo UNet(4,1) {tuple(x.shape)}: logits rel err vs the reference's own forward {e:.2e}")
    assert got.shape == want.shape and e < TOL


def test_simple_unet_other_shapes_and_contract(hip):
    from oracle.unet_ref import simple_unet_ref
    from starcop_amd.unet_simple import SimpleUNet
    net = SimpleUNet(4, 1)
    net.load_state_dict(g9_util.full_unet_state())
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 128, 200, generator=g)
    with torch.no_grad():
        want = simple_unet_ref(g9_util.full_unet_state(), x)
    got = net(x.to(DEV)).cpu()
    assert float((got - want).abs().max() / want.abs().max()) < TOL
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 4, 60, 64, device=DEV))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 64, 64, device=DEV))


def test_g9_full_unet_backward_hip(hip):
    """the whole in-repo UNet BACKWARD on the HIP kernels (split-MFMA data / weight gradients with two-source concat inputs and
    channel-split outputs, bias gradients from sc_bn_bwd_reduce, sc_maxpool2x2_bwd, sc_upsample_bilinear2x_bwd) against the gradients
    the REFERENCE's autograd produced for the same weights, input and upstream gradient: every bias gradient and both marginal sums
    of every filter gradient, 15 convolutions deep.  Gate 1e-3 (SURVEY 8d gradient gate); the worst entries are printed."""
    from starcop_amd.unet_simple import SimpleUNet
    net = SimpleUNet(4, 1)
    net.load_state_dict(g9_util.full_unet_state())
    net = net.to(DEV).train()
    x, r = g9_util.full_unet_grad_case()
    y = net(x.to(DEV))
    (y * r.to(DEV)).sum().backward()
    z = g9_util.load()
    errs = g9_util.full_unet_grad_errs(z, {k: p.grad for k, p in net.named_parameters()})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("in-repo UNet backward vs the reference's autograd: worst " + ", ".join(f"{k} {v:.2e}" for k, v in worst))
    assert len(errs) == 45 and worst[0][1] < 1e-3, worst
    # forward of the training path equals the inference path
    net.eval()
    assert torch.equal(net(x.to(DEV)), y.detach())
