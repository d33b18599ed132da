"""One-launch INFERENCE execution of a MobileNetV2 inverted-residual block (csrc/conv_irb.hip: sc_irb_eval) against a float64
evaluation of the torch ops the reference dispatches for it in eval mode -- F.conv2d 1x1, eval-mode batch_norm (a per-channel affine),
relu6, depthwise F.conv2d, F.conv2d 1x1 and the residual add (torchvision InvertedResidual inside smp.Unet('mobilenet_v2'),
/root/reference/starcop/models/model_module.py:244-251; the eval path of model_module.py:90-98 / utils/padding.py:13-50).  Through the
C ABI; filters packed by the same batched pack launch the network uses."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_ops import DEV, cst_affine, dev, pack_pw3, relerr  # noqa: E402
from starcop_amd import _lib  # noqa: E402
from starcop_amd._lib import ACT_NONE, SRC_AFFINE, SRC_RAW, check, make_src, ptr, sc_irb_args, stream  # noqa: E402

# (N, Cin, hidden, Cout, H, W, residual, input source[, stride]): features.8-13 (32 x 32 planes: 8 x 8 tiles, 32-channel chunks, one / two
# projection pairs per wave), features.15-17 (16 x 16: 4 x 8 tiles, 64-channel chunks, two / three pairs), ragged planes (the
# 40 x 39 planes of a 1280 x 1248 scene; H, W not multiples of the tile, W % 4 != 0: the scalar store path), channel counts that
# are not multiples of 16 / 32, hidden % 64 != 0 on a small plane (falls back to the 8 x 8 tiling)
CASES = [
    (2, 64, 384, 64, 32, 32, True, "raw"),
    (2, 64, 384, 96, 32, 32, False, "affine"),
    (1, 96, 576, 96, 32, 32, True, "affine"),
    (2, 160, 960, 160, 16, 16, True, "raw"),
    (1, 160, 960, 320, 16, 16, False, "raw"),
    (1, 64, 384, 64, 40, 39, True, "affine"),
    (1, 24, 96, 40, 13, 11, False, "raw"),
    (3, 32, 192, 32, 20, 12, True, "raw"),
    (2, 96, 576, 96, 8, 8, True, "affine"),
    # stride 2 (features.7: 32 -> 192 -> 64 from 64 x 64; features.14: 96 -> 576 -> 160 from 32 x 32; odd input planes: the 39-wide planes
    # of a scene, output (H - 1) / 2 + 1)
    (2, 32, 192, 64, 64, 64, False, "raw", 2),
    (2, 96, 576, 160, 32, 32, False, "affine", 2),
    (1, 32, 192, 64, 40, 39, False, "affine", 2),
    (1, 64, 128, 24, 9, 13, False, "raw", 2),
]


def _run(case, seed):
    N, Cin, hid, Cout, H, W, res, mode = case[:8]
    stride = case[8] if len(case) > 8 else 1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    lib = _lib.load()
    assert lib.sc_irb_supported(Cin, hid, Cout, H, W, stride)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g) * 1.5
    xs, xh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    We = torch.randn(hid, Cin, generator=g) * (2.0 / Cin) ** 0.5
    Wd = torch.randn(hid, 3, 3, generator=g) * 0.4
    Wp = torch.randn(Cout, hid, generator=g) * (1.0 / hid) ** 0.5
    se, he = torch.rand(hid, generator=g) + 0.5, torch.randn(hid, generator=g) * 0.5
    sd, hd = torch.rand(hid, generator=g) + 0.5, torch.randn(hid, generator=g) * 0.5
    sp, hp = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.5
    affine = mode == "affine"
    # float64 reference
    xa = x.double() * xs.double()[None, :, None, None] + xh.double()[None, :, None, None] if affine else x.double()
    e = F.relu6(F.conv2d(xa, We.double()[:, :, None, None]) * se.double()[None, :, None, None] + he.double()[None, :, None, None])
    d = F.relu6(F.conv2d(e, Wd.double()[:, None], stride=stride, padding=1, groups=hid) * sd.double()[None, :, None, None] + hd.double()[None, :, None, None])
    p = F.conv2d(d, Wp.double()[:, :, None, None])
    want = xa + p * sp.double()[None, :, None, None] + hp.double()[None, :, None, None] if res else p
    # HIP
    xd = dev(x)
    a = sc_irb_args()
    a.x = make_src(xd, Cin, SRC_AFFINE, act=ACT_NONE, cst=cst_affine(xs, xh)) if affine else make_src(xd, Cin, SRC_RAW)
    wpe, wpp = pack_pw3(dev(We[:, :, None, None]), 0), pack_pw3(dev(Wp[:, :, None, None]), 0)
    ce, cd, cp = cst_affine(se, he), cst_affine(sd, hd), cst_affine(sp, hp)
    wd = dev(Wd)
    out = torch.full((N, Cout, Ho, Wo), float("nan"), device=DEV)
    zmax = torch.zeros(1, device=DEV)
    a.wpk_expand, a.cst_expand, a.w_dw, a.cst_dw = wpe.data_ptr(), ce.data_ptr(), wd.data_ptr(), cd.data_ptr()
    a.wpk_project, a.cst_project = wpp.data_ptr(), (cp.data_ptr() if res else None)
    a.out, a.z_absmax = out.data_ptr(), (zmax.data_ptr() if res else None)
    a.N, a.Cin, a.hidden, a.Cout, a.H, a.W, a.stride, a.residual = N, Cin, hid, Cout, H, W, stride, int(res)
    check(lib.sc_irb_eval(C.byref(a), stream()))
    torch.cuda.synchronize()
    return out, want, zmax


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_inverted_residual_block_eval_one_launch(hip, case):
    out, want, zmax = _run(case, seed=100 + CASES.index(case))
    assert bool(torch.isfinite(out).all())
    assert relerr(out, want) < 3e-6, relerr(out, want)
    if case[6]:      # the range record of the residual sum: max |z| of this launch (fp32 rounding of the same values)
        assert abs(float(zmax) - float(out.abs().max())) <= 1e-6 * float(out.abs().max())


def test_inverted_residual_block_eval_rejects_what_it_does_not_take(hip):
    lib = _lib.load()
    assert not lib.sc_irb_supported(160, 960, 320, 16, 16, 2)    # stride 2: Cin <= 96 only
    assert not lib.sc_irb_supported(64, 384, 64, 32, 32, 3)
    assert not lib.sc_irb_supported(64, 100, 64, 32, 32, 1)      # hidden % 32
    assert not lib.sc_irb_supported(320, 1280, 64, 32, 32, 1)    # Cin > 160: the patch would not fit the LDS
    a = sc_irb_args()
    a.stride = 3
    assert lib.sc_irb_eval(C.byref(a), stream()) != 0


def test_network_eval_with_fused_blocks_equals_the_separate_launches_and_the_oracle(hip, monkeypatch):
    """whole network, eval mode: every supported inverted-residual block as one launch (STARCOP_IRB=all semantics) against the three
    / four separate launches and against the CPU oracle; the range record of the residual sums must come out the same; and an eval
    forward right after a TRAINING step (the validation loop) must see freshly packed filters (the fused blocks' PW3 layouts are
    packed by inference forwards only)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_unet import make_pair, ref_normalize, synth_batch, to_dev
    from starcop_amd import network as nw
    batch = synth_batch(2, 128, 160, seed=5)
    outs, recs = {}, {}
    for mode in ("all", "0"):
        monkeypatch.setattr(nw, "_IRB", mode)
        model, ref = make_pair(seed=3)
        model.eval(); ref.eval()
        with torch.no_grad():
            outs[mode] = model(to_dev(batch)["input"]).clone()
            want = ref(ref_normalize(batch["input"]))
        plan = model.network._plans[(2, 128, 160)]
        assert (len(plan.irb) > 0) == (mode == "all")
        recs[mode] = plan.fin_amax.clone()
        assert relerr(outs[mode], want) < 1e-4, (mode, relerr(outs[mode], want))
    assert relerr(outs["all"], outs["0"]) < 2e-5
    assert torch.allclose(recs["all"], recs["0"], rtol=1e-5, atol=0)
    # validation after a training step: parameters changed through the flat buffer, the eval forward must repack its layouts
    monkeypatch.setattr(nw, "_IRB", "all")
    model, ref = make_pair(seed=4)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    model.fused_train_step(to_dev(batch), opt)
    ref.load_state_dict({k: v.cpu() for k, v in model.network.state_dict().items()})
    model.eval(); ref.eval()
    with torch.no_grad():
        got = model(to_dev(batch)["input"])
        want = ref(ref_normalize(batch["input"]))
    assert len(model.network._plans[(2, 128, 160)].irb) > 0
    assert relerr(got, want) < 1e-4, relerr(got, want)
    model.train()
    model.fused_train_step(to_dev(batch), opt)          # ... and training continues on the separate kernels
    ref.load_state_dict({k: v.cpu() for k, v in model.network.state_dict().items()})
    model.eval()
    with torch.no_grad():
        assert relerr(model(to_dev(batch)["input"]), ref(ref_normalize(batch["input"]))) < 1e-4
