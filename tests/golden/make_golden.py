"""Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; it cannot travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules on the hot path are imported unmodified; third-party packages that are absent
from the image and only needed at import time (rasterio, spectral, wandb, pytorch_lightning,
torchmetrics, ...) are replaced by empty stub modules.  ``spectral.io.envi.open`` is stubbed by a small
ENVI-BSQ reader so ``generate_template_from_bands`` can read its ch4.lut.  Only DATA is written: inputs,
seeds and the reference's outputs (.npz); no reference source text is stored.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Lut:
    """spectral.io.envi.open(hdr, lut) stand-in: ENVI BSQ float64 LE, samples x lines x bands from the header."""

    def __init__(self, hdr, dat):
        txt = open(hdr).read()
        import re
        get = lambda k: int(re.search(rf"{k}\s*=\s*(\d+)", txt).group(1))   # noqa: E731
        ns, nl, nb = get("samples"), get("lines"), get("bands")
        wl = re.search(r"wavelength\s*=\s*\{([^}]*)\}", txt, re.S).group(1)
        self.bands = types.SimpleNamespace(centers=[float(v) for v in wl.replace("\n", " ").split(",") if v.strip()])
        raw = np.fromfile(dat, dtype="<f8")
        self._arr = raw.reshape(nb, nl, ns).transpose(1, 2, 0)    # (lines, samples, bands)

    def asarray(self):
        return self._arr


_stub("rasterio")
envi = types.SimpleNamespace(open=lambda hdr, dat=None: _Lut(hdr, dat))
_stub("spectral", io=types.SimpleNamespace(envi=envi))

import torch  # noqa: E402
from starcop.models import mag1c as ref_mag1c  # noqa: E402
from starcop.data import normalizer_module as ref_norm  # noqa: E402
from starcop.data import feature_extration as ref_feat  # noqa: E402
from starcop.models.utils import padding as ref_pad  # noqa: E402
from starcop import metrics as ref_metrics  # noqa: E402

torch.set_num_threads(4)


def synth_group(rng, P, S, templ, dtype, plume_frac=0.2):
    base = rng.uniform(1.0, 6.0, size=S)
    x = base[None, :] * (1.0 + 0.05 * rng.standard_normal((P, S)))
    k = np.zeros(P)
    idx = rng.choice(P, int(P * plume_frac), replace=False)
    k[idx] = rng.uniform(0.2e-5, 5e-5, size=idx.size)
    x = x * (1.0 + k[:, None] * templ[None, :])      # template is negative (absorption)
    return x.astype(dtype)


def g3_templates():
    out = {}
    aviris_c = np.array(ref_feat.AVIRIS_WAVELENGTHS, dtype=np.float64)
    aviris_f = np.full_like(aviris_c, 5.6)
    t = ref_mag1c.generate_template_from_bands(aviris_c, aviris_f)
    keep = ref_mag1c.get_mask_bad_bands(aviris_c) & (aviris_c >= 2122) & (aviris_c <= 2488)
    out["aviris_centers"], out["aviris_fwhm"], out["aviris_keep"] = aviris_c, aviris_f, keep
    out["aviris_template_kept"] = t[keep]                    # (73, 2): only in-LUT-range rows are meaningful
    emit_c = np.linspace(381.0, 2493.0, 285)
    emit_f = np.full_like(emit_c, 8.5)
    te = ref_mag1c.generate_template_from_bands(emit_c, emit_f)
    ek = (emit_c >= 2122) & (emit_c <= 2488)
    out["emit_centers"], out["emit_fwhm"], out["emit_keep"] = emit_c, emit_f, ek
    out["emit_template_kept"] = te[ek]
    wave = np.array([350., 400., 1349., 1350., 1351., 1420., 1421., 1800., 1801., 1944., 1945., 2485., 2486.])
    out["badband_wave"], out["badband_keep"] = wave, ref_mag1c.get_mask_bad_bands(wave)
    np.savez_compressed(os.path.join(OUT, "g3_templates.npz"), **out)
    return out


def g1_filters(tpl):
    rng = np.random.default_rng(1234)
    templ73 = tpl["aviris_template_kept"][:, 1]
    templ24 = templ73[::3][:24].copy()
    cases = {}

    def run(name, x, templ, fn, **kw):
        xt, tt = torch.tensor(x), torch.tensor(templ.astype(x.dtype))
        kw_t = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        mf, R = fn(xt, tt, **kw_t)[:2]
        cases[name + "_x"], cases[name + "_t"] = x, templ.astype(x.dtype)
        cases[name + "_mf"] = mf.numpy()
        cases[name + "_R"] = R.numpy() if torch.is_tensor(R) else np.asarray(R)
        for k, v in kw.items():
            cases[f"{name}_kw_{k}"] = np.asarray(v)

    for dt in (np.float64, np.float32):
        tag = "f64" if dt == np.float64 else "f32"
        xa = np.stack([synth_group(rng, 300, 24, templ24, dt) for _ in range(2)])
        xb = synth_group(rng, 512, 73, templ73, dt)[None]
        for alpha in (0.0, 1e-4):
            at = "a0" if alpha == 0 else "a1e4"
            run(f"rmf_{tag}_{at}_b2", xa, templ24, ref_mag1c.rmf, alpha=alpha)
            run(f"acr_{tag}_{at}_b2", xa, templ24, ref_mag1c.acrwl1mf, num_iter=30, alpha=alpha)
            run(f"acr_{tag}_{at}_p512", xb, templ73, ref_mag1c.acrwl1mf, num_iter=30, alpha=alpha)
        mask = rng.uniform(size=300) > 0.3
        run(f"rmf_{tag}_mask", xa, templ24, ref_mag1c.rmf, alpha=1e-4, mask=mask)
        run(f"acr_{tag}_mask", xa, templ24, ref_mag1c.acrwl1mf, num_iter=30, alpha=1e-4, mask=mask)
        run(f"acr_{tag}_albedo", xa, templ24, ref_mag1c.acrwl1mf, num_iter=10, albedo_override=True)
        run(f"acr_{tag}_sparse", xa, templ24, ref_mag1c.acrwl1mf, num_iter=10, sparse_override=True)
        run(f"acr_{tag}_cus", xa, templ24, ref_mag1c.acrwl1mf, num_iter=10, covariance_update_scaling=0.5)
        run(f"rmf_{tag}_noscale_zero", xa, templ24, ref_mag1c.rmf, zero_override=True, apply_scaling=False)
    # fewer pixels than bands with alpha=0: singular covariance -> the reference raises LinAlgError (mag1c.py:251,323)
    xs = synth_group(rng, 16, 24, templ24, np.float64)[None]
    try:
        ref_mag1c.acrwl1mf(torch.tensor(xs), torch.tensor(templ24), num_iter=2, alpha=0.0)
        raised = False
    except torch.linalg.LinAlgError:
        raised = True
    cases["singular_x"], cases["singular_t"], cases["singular_raises"] = xs, templ24, np.array(raised)
    np.savez_compressed(os.path.join(OUT, "g1_filters.npz"), **cases)


def g2_groups(tpl):
    rng = np.random.default_rng(77)
    templ = tpl["aviris_template_kept"][:, 1][::3][:24].copy()
    H, W, S = 64, 48, 24
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cube = synth_group(rng, H * W, S, templ, dt).reshape(H, W, S)
        groups = (np.arange(W)[None, :] // 2 + 3).repeat(H, 0).astype(np.int64)
        groups[:7, 40:42] = 99          # a tiny group: 14 px ...
        groups[:2, 40:42] = 98          # ... and one of 4 px (<= 10 -> skipped, stays NODATA)
        cube[5, 5, :] = ref_mag1c.NODATA  # invalid pixel
        cube[20:23, 10, 3] = ref_mag1c.NODATA
        spec = torch.tensor(templ.astype(dt))
        fn = lambda x: ref_mag1c.acrwl1mf(torch.as_tensor(x), spec, num_iter=30, alpha=1e-4)   # noqa: E731
        mf, alb = ref_mag1c.func_by_groups(fn, cube.copy(), groups, mask=None, disable_pbar=True, samples_read=5)
        valid = groups != 0
        valid[40:44, :] = False
        mf2, alb2 = ref_mag1c.func_by_groups(fn, cube.copy().clip(0.1, None), groups, mask=valid, disable_pbar=True)
        np.savez_compressed(os.path.join(OUT, f"g2_groups_{tag}.npz"), cube=cube, groups=groups, templ=templ.astype(dt),
                            mf=mf.numpy(), albedo=alb.numpy(), mask2=valid, mf2=mf2.numpy(), albedo2=alb2.numpy())


def g4_normalizer():
    S = lambda **k: types.SimpleNamespace(**k)   # noqa: E731
    rng = np.random.default_rng(5)
    out = {}
    for tag, prods in (("cfg4", ["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"]),
                       ("ratio", ["ratio_aviris_2350_2310_out", "ratio_wv3_B8_B8MLR_SanchezGarcia22_sum_c_out",
                                  "ratio_wv3_B8_B8MLR_SanchezGarcia22_simplediv", "not_a_product"])):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            n = ref_norm.DataNormalizer(S(dataset=S(input_products=prods, output_products=["labelbinary"])))
        scale = np.array([3000., 100., 100., 100.]) if tag == "cfg4" else np.array([0.2, 0.3, 3.0, 20.0])
        x = (rng.standard_normal((2, 4, 9, 11)) * scale[None, :, None, None]).astype(np.float32)
        y = (rng.uniform(size=(2, 1, 9, 11)) > 0.5).astype(np.float32)
        out[f"{tag}_x"], out[f"{tag}_y"] = x, y
        out[f"{tag}_xn"] = n.normalize_x(torch.tensor(x)).numpy()
        out[f"{tag}_yn"] = n.normalize_y(torch.tensor(y)).numpy()
        out[f"{tag}_xdn"] = n.denormalize_x(torch.tensor(x)).numpy()
        out[f"{tag}_products"] = np.array(prods)
        out[f"{tag}_param_dtype"] = np.array(str(n.factors_input.dtype))
    np.savez_compressed(os.path.join(OUT, "g4_normalizer.npz"), **out)


def g5_masks():
    """pred_classification / differences: the reference functions live in model_module.py, whose imports
    (wandb, pytorch_lightning, torchmetrics, starcop.utils -> rasterio/fsspec) are stubbed."""
    _stub("wandb")
    _stub("pytorch_lightning", LightningModule=torch.nn.Module)
    tm = _stub("torchmetrics")
    tm.functional = _stub("torchmetrics.functional", mean_squared_error=None, mean_absolute_error=None)
    _stub("starcop.utils", get_filesystem=lambda p: None)
    from starcop.models import model_module as ref_mm
    rng = np.random.default_rng(9)
    out = {}
    for tag, (H, W) in (("64", (64, 64)), ("128", (128, 128)), ("96x160", (96, 160))):
        thr = int(10 * H * W / 64 ** 2)
        pb = np.zeros((4, 1, H, W), dtype=np.int64)
        pb[0].reshape(-1)[:thr] = 1            # == threshold -> 0
        pb[1].reshape(-1)[:thr + 1] = 1        # threshold + 1 -> 1
        pb[2] = (rng.uniform(size=(1, H, W)) > 0.5)
        gt = (rng.uniform(size=(4, 1, H, W)) > 0.7).astype(np.int64)
        out[f"pb_{tag}"], out[f"gt_{tag}"] = pb, gt
        out[f"cls_{tag}"] = ref_mm.pred_classification(torch.tensor(pb)).numpy()
        out[f"diff_{tag}"] = ref_mm.differences(torch.tensor(pb), torch.tensor(gt)).numpy()
    z = np.array([-3.0, -1e-8, -0.0, 0.0, 1e-8, 1e-7, 2e-7, 1e-3, 4.0], dtype=np.float32)
    out["tie_logits"] = z
    out["tie_ge0"] = (torch.tensor(z) >= 0).long().numpy()
    out["tie_sig"] = (torch.sigmoid(torch.tensor(z)) > .5).long().numpy()
    np.savez_compressed(os.path.join(OUT, "g5_masks.npz"), **out)


def g6_padding():
    vs = np.array([1, 31, 32, 33, 64, 70, 90, 512, 1242, 1280, 2007])
    out = {"v": vs, "pad32": np.array([ref_pad.find_padding(int(v), 32) for v in vs]),
           "pad8": np.array([ref_pad.find_padding(int(v)) for v in vs])}
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 45, 70)).astype(np.float32)

    class Smooth(torch.nn.Module):      # identity-like model with a receptive field: 3x3 box filter of channel sum
        def forward(self, t):
            return torch.nn.functional.avg_pool2d(t.sum(1, keepdim=True), 3, 1, 1, count_include_pad=False)

    out["pp_x"] = x
    out["pp_out3d"] = ref_pad.padded_predict(x, Smooth(), 32)
    out["pp_out2d"] = ref_pad.padded_predict(x, lambda t: Smooth()(t)[:, 0], 32)
    np.savez_compressed(os.path.join(OUT, "g6_padding.npz"), **out)


def g7_metrics():
    cms = np.array([[[90, 10], [5, 20]], [[1000, 3], [7, 40]], [[5, 0], [0, 9]], [[50, 50], [50, 50]]], dtype=np.int64)
    out = {"cm": cms}
    for fn in ref_metrics.METRICS_CONFUSION_MATRIX + [ref_metrics.TP, ref_metrics.TN, ref_metrics.FP, ref_metrics.FN, ref_metrics.FPR]:
        out[fn.__name__] = np.array([float(fn(torch.tensor(c))) for c in cms])
    np.savez_compressed(os.path.join(OUT, "g7_metrics.npz"), **out)


def g8_ratio():
    rng = np.random.default_rng(21)
    sig = rng.uniform(0.5, 3.0, size=(1, 64, 64)).astype(np.float32)
    bg = (sig * rng.uniform(0.9, 1.2, size=sig.shape)).astype(np.float32)
    sig[0, :4, :4] = 0.0; bg[0, :4, :4] = 0.0             # zero/zero pixels -> -0.6
    sig[0, 10, 10] = 50.0                                  # outlier
    out = {"sig": sig, "bg": bg, "ratio": ref_feat.ratio_2c_match_c_from_sums_outlier(bg.copy(), sig.copy())}
    m = rng.uniform(-100, 1500, size=(32, 32)).astype(np.float32)
    out["mag1c"], out["weight"] = m, ref_feat.weight_mag1c(m)
    np.savez_compressed(os.path.join(OUT, "g8_ratio.npz"), **out)


# ---- G9: the reference's own convolution blocks (the only conv arithmetic the reference itself holds) -------------------------
# Weights, biases, inputs and the upstream gradient are drawn from numpy PCG64 *uniform* streams (bit-stable across numpy
# versions), so tests/g9_util.py regenerates them from the seeds below and only the reference's OUTPUTS are stored
# (strided crops + fp64 per-channel checksums for the big tensors).
G9_CASES = (  # name, how the block is obtained from the reference, input shape, seed
    ("dc_4_8", ("double_conv", 4, 8), (1, 4, 64, 64), 901),
    ("unet_down1", ("UNet", "dconv_down1"), (1, 4, 64, 64), 902),        # 4 -> 64 -> 64
    ("unet_down2", ("UNet", "dconv_down2"), (1, 64, 32, 32), 903),       # 64 -> 128 -> 128
    ("unet_up1", ("UNet", "dconv_up1"), (1, 192, 64, 64), 904),          # cat(128, 64) -> 64 -> 64
    ("unet_last", ("UNet", "conv_last"), (1, 64, 64, 64), 905),          # 1x1, 64 -> 1, bias
)


def g9_fill(rng, shape, bound):
    return ((rng.random(shape) * 2.0 - 1.0) * bound).astype(np.float32)


def g9_params(seed, convs):
    """convs: [(cout, cin, k)] -> [(weight, bias)] drawn in order from PCG64(seed): U(-1/sqrt(fan_in), ..) and U(-0.1, 0.1)"""
    rng = np.random.default_rng(seed)
    return [(g9_fill(rng, (co, ci, k, k), 1.0 / np.sqrt(ci * k * k)), g9_fill(rng, (co,), 0.1)) for co, ci, k in convs]


def g9_crop(a):
    """what is stored of a tensor: everything if small; else a stride-3 lattice over the image axes (filters: every second
    output and input channel, all taps) + fp64 checksums over the axes that were thinned"""
    a = np.asarray(a)
    if a.size <= 40000:
        return {"full": a}
    if a.shape[-1] <= 3:
        return {"crop2": a[::2, ::2].copy(), "cisum": a.astype(np.float64).sum(axis=1)}
    return {"crop3": a[..., ::3, ::3].copy(), "chsum": a.astype(np.float64).sum(axis=(-2, -1))}


G9_RELU_MARGIN = 1e-5
G9_FULL_SEED = 4242
G9_SEED_TRIES = 4000


def g9_convblocks():
    """A ReLU is a discrete switch: a pre-activation within fp32 rounding of zero may legitimately fall on either side in two
    correct implementations and then moves a whole bias/filter gradient by O(|dL/dy|).  The seed of every case is therefore
    advanced until no pre-activation of the block lies within G9_RELU_MARGIN of zero (10x the rounding noise of a K <= 1728
    fp32 reduction of O(1) terms); the chosen seed is stored in <name>.meta[0]."""
    from starcop.models.architectures import layer_factory as ref_lf, unet as ref_unet
    torch.manual_seed(0)
    net = ref_unet.UNet(4, 1)
    out = {}
    for name, how, xshape, seed0 in G9_CASES:
        block = ref_lf.double_conv(how[1], how[2]) if how[0] == "double_conv" else getattr(net, how[1])
        convs = [m for m in ([block] if isinstance(block, torch.nn.Conv2d) else block) if isinstance(m, torch.nn.Conv2d)]
        pre = []
        hooks = [c.register_forward_hook(lambda m, i, o: pre.append(float(o.detach().abs().min()))) for c in convs]
        best = (-1.0, -1)
        for seed in list(range(seed0, seed0 + 7 * G9_SEED_TRIES, 7)) + [None]:
            if seed is None:                       # no seed reached the margin: take the best one seen
                seed = best[1]
            params = g9_params(seed, [(c.out_channels, c.in_channels, c.kernel_size[0]) for c in convs])
            with torch.no_grad():
                for c, (w, b) in zip(convs, params):
                    c.weight.copy_(torch.from_numpy(w)); c.bias.copy_(torch.from_numpy(b))
            rng = np.random.default_rng(seed + 1000)
            x = torch.from_numpy(g9_fill(rng, xshape, 1.5)).requires_grad_(True)
            del pre[:]
            for c in convs:
                c.zero_grad()
            y = block(x)
            if how[1] == "conv_last" or min(pre) > G9_RELU_MARGIN or seed == best[1]:
                break
            best = max(best, (min(pre), seed))
        for h in hooks:
            h.remove()
        print(f"G9 {name}: seed {seed}, min |pre-activation| {min(pre):.2e}")
        r = torch.from_numpy(g9_fill(rng, tuple(y.shape), 1.0))
        (y * r).sum().backward()
        rec = {"y": y.detach().numpy(), "gx": x.grad.numpy()}
        for i, c in enumerate(convs):
            rec[f"gw{i}"], rec[f"gb{i}"] = c.weight.grad.numpy(), c.bias.grad.numpy()
        for k, v in rec.items():
            for kk, vv in g9_crop(v).items():
                out[f"{name}.{k}.{kk}"] = vv
        out[f"{name}.meta"] = np.array([seed] + list(xshape) + [c.out_channels for c in convs], dtype=np.int64)
    # ---- the whole in-repo UNet(4, 1): 7.78 M parameters drawn from one PCG64 stream in state_dict order (weights U(+-sqrt(3/fan_in)),
    # i.e. variance-preserving under ReLU would need 6/fan_in; 3/fan_in keeps activations O(1) through 15 layers; biases U(+-0.1))
    torch.manual_seed(0)
    full = ref_unet.UNet(4, 1).eval()
    rng = np.random.default_rng(G9_FULL_SEED)
    with torch.no_grad():
        for k, v in full.state_dict().items():
            if k.endswith("weight"):
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                v.copy_(torch.from_numpy(g9_fill(rng, tuple(v.shape), np.sqrt(6.0 / fan_in))))
            else:
                v.copy_(torch.from_numpy(g9_fill(rng, tuple(v.shape), 0.1)))
        for tag, shape in (("a", (1, 4, 64, 64)), ("b", (2, 4, 96, 64))):
            x = torch.from_numpy(g9_fill(np.random.default_rng(G9_FULL_SEED + 1 + len(tag) + shape[0]), shape, 1.5))
            out[f"unet_full.{tag}.y"] = full(x).numpy()
            out[f"unet_full.{tag}.shape"] = np.array(shape, dtype=np.int64)
    # backward of the whole network (conv / ReLU / MaxPool2d / bilinear-upsample autograd, 15 convolutions deep): L = sum(y * r).
    # A ReLU input within rounding of zero, or a max-pool window whose two largest values are within rounding of each other, is a
    # discrete switch that two correct fp32 implementations may set differently -- and ONE flipped pixel moves every upstream
    # gradient sum by ~1 % (sums of random-sign terms are only sqrt(n) larger than a term).  The case is therefore small
    # (2 x 4 x 16 x 16: ~0.25 M activations) and its input seed is advanced until every ReLU input is further than
    # G9_RELU_MARGIN from zero and every positive pooling window has a top-2 gap above it.
    # stored per parameter: bias gradients in full, filter gradients as their two marginal sums (over cin and over cout, fp64)
    margins = []
    hooks = [m.register_forward_hook(lambda mod, i, o: margins.append(float(o.detach().abs().min())))
             for n_, m in full.named_modules() if isinstance(m, torch.nn.Conv2d) and n_ != "conv_last"]

    def pool_gap(mod, i, o):
        t = torch.nn.functional.unfold(i[0].detach(), 2, stride=2).reshape(i[0].shape[0], i[0].shape[1], 4, -1)
        top = t.topk(2, dim=2).values
        pos = top[:, :, 0] > 0
        if bool(pos.any()):
            margins.append(float((top[:, :, 0] - top[:, :, 1])[pos].min()))
    hooks.append(full.maxpool.register_forward_hook(pool_gap))
    best = (-1.0, -1)
    gshape = (2, 4, 16, 16)
    for seed in list(range(G9_FULL_SEED + 100, G9_FULL_SEED + 100 + 3000)) + [None]:
        if seed is None:
            seed = best[1]
        x = torch.from_numpy(g9_fill(np.random.default_rng(seed), gshape, 1.5))
        del margins[:]
        with torch.no_grad():
            full(x)
        if min(margins) > G9_RELU_MARGIN or seed == best[1]:
            break
        best = max(best, (min(margins), seed))
    print(f"G9 unet_full backward case: input seed {seed}, smallest switch margin {min(margins):.2e}")
    for h in hooks:
        h.remove()
    y = full(x)
    r = torch.from_numpy(g9_fill(np.random.default_rng(seed + 77), tuple(y.shape), 1.0))
    full.zero_grad()
    (y * r).sum().backward()
    out["unet_full.grad.meta"] = np.array([seed] + list(gshape), dtype=np.int64)
    for k, p_ in full.named_parameters():
        gk = p_.grad.numpy().astype(np.float64)
        if gk.ndim == 1:
            out[f"unet_full.grad.{k}"] = gk
        else:
            out[f"unet_full.grad.{k}.cisum"] = gk.sum(axis=1)
            out[f"unet_full.grad.{k}.cosum"] = gk.sum(axis=0)
    np.savez_compressed(os.path.join(OUT, "g9_convblocks.npz"), **out)


# ---- G10: the EMIT driver itself (starcop/models/mag1c_emit.py:16-90) on a duck-typed raster ---------------------------------
def g10_emit_cube(seed, rows, cols, centers):
    """(rows, cols, S) float32 radiance-like cube from PCG64 uniform streams (regenerated by tests/g9_util.py::emit_cube)"""
    rng = np.random.default_rng(seed)
    S = centers.size
    base = 1.0 + 5.0 * rng.random(S)
    raw = (base * (1.0 + 0.1 * (rng.random((rows, cols, S)) - 0.5))).astype(np.float32)
    dip = np.exp(-0.5 * ((centers - 2300.0) / 60.0) ** 2)                # a plume-like absorption around 2300 nm
    raw[20:40, 3:6, :] *= (1.0 - 0.03 * dip).astype(np.float32)
    raw[:7, :3, :] = -9999.0                                             # fill wedge
    raw[50, 7, 250] = -9999.0                                            # a single fill sample inside the mag1c band range
    return raw


def g10_emit_driver():
    geo = _stub("georeader"); rd = _stub("georeader.readers"); em = _stub("georeader.readers.emit", EMITImage=object)
    gt = _stub("georeader.geotensor", GeoTensor=object)
    geo.readers, rd.emit, geo.geotensor = rd, em, gt
    from starcop.models import mag1c_emit as ref_emit
    g3 = np.load(os.path.join(OUT, "g3_templates.npz"))

    class EI:                                      # what mag1c_emit touches of georeader's EMITImage
        fill_value_default = -9999.0

        def __init__(self, raw, wavelengths, fwhm):
            self.raw, self.wavelengths, self.fwhm = raw, wavelengths, fwhm

        def read_from_bands(self, sel):
            return EI(self.raw[..., sel], self.wavelengths[sel], self.fwhm[sel])

        def load_raw(self, transpose=False):
            return self.raw
    seed, rows, cols = 77, 96, 10
    raw = g10_emit_cube(seed, rows, cols, g3["emit_centers"])
    out = {"meta": np.array([seed, rows, cols], dtype=np.int64)}
    for step in (2, 4, None):
        mf, alb = ref_emit.mag1c_emit(EI(raw, g3["emit_centers"], g3["emit_fwhm"]), column_step=step, georreferenced=False, display_pbar=False)
        out[f"mf_step{step}"], out[f"albedo_step{step}"] = mf, alb
    np.savez_compressed(os.path.join(OUT, "g10_emit_driver.npz"), **out)


def g11_energy():
    """compute_energy=True (mag1c.py:270-275, 337-343) of rmf / acrwl1mf: the reference's own numbers for float64 and float32 groups,
    with and without a statistics mask, alpha = 0 and 1e-4.  Without a mask the residual term is the sum of a P x P matrix whose exact
    value is zero (rounding noise of the reference's arithmetic): those cases pin the log-det term and the iteration terms."""
    g3 = np.load(os.path.join(OUT, "g3_templates.npz"))
    rng = np.random.default_rng(4321)
    templ24 = g3["aviris_template_kept"][:, 1][::3][:24].copy()
    out = {}
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        xa = np.stack([synth_group(rng, 200, 24, templ24, dt) for _ in range(2)])
        mask = rng.uniform(size=200) > 0.3
        out[f"x_{tag}"], out[f"mask_{tag}"] = xa, mask
        for alpha, at in ((0.0, "a0"), (1e-4, "a1e4")):
            for mk, mt in ((None, "nomask"), (mask, "mask")):
                kw = {} if mk is None else {"mask": torch.tensor(mk)}
                mf, R, e = ref_mag1c.rmf(torch.tensor(xa), torch.tensor(templ24.astype(dt)), alpha=alpha, compute_energy=True, **kw)
                out[f"rmf_{tag}_{at}_{mt}_mf"], out[f"rmf_{tag}_{at}_{mt}_e"] = mf.numpy(), np.asarray(e, dtype=np.float64)
                mf, R, el = ref_mag1c.acrwl1mf(torch.tensor(xa), torch.tensor(templ24.astype(dt)), num_iter=5, alpha=alpha,
                                               compute_energy=True, **kw)
                out[f"acr_{tag}_{at}_{mt}_mf"] = mf.numpy()
                out[f"acr_{tag}_{at}_{mt}_e0"] = np.asarray(el[0], dtype=np.float64)
                out[f"acr_{tag}_{at}_{mt}_e"] = np.array([float(v) for v in el[1:]], dtype=np.float64)
    out["t"] = templ24
    np.savez_compressed(os.path.join(OUT, "g11_energy.npz"), **out)


if __name__ == "__main__":
    tpl = g3_templates()
    g1_filters(tpl)
    g2_groups(tpl)
    g4_normalizer()
    g6_padding()
    g7_metrics()
    g8_ratio()
    g5_masks()
    g9_convblocks()
    g10_emit_driver()
    g11_energy()
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} bytes")
