"""G9 golden vectors: the reference's own convolution blocks (layer_factory.double_conv, architectures/unet.py UNet
sub-blocks; /root/reference/starcop/models/architectures/layer_factory.py:4-9, unet.py:7-51), run by the reference in the
build container (tests/golden/make_golden.py::g9_convblocks).  Inputs, filters, biases and the upstream gradient are
regenerated here from the same numpy PCG64 uniform streams; only the reference's outputs are stored in g9_convblocks.npz.
"""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_convblocks.npz")

# name -> (conv list [(cout, cin, k)], input shape); activation after every conv of a double_conv is ReLU, conv_last has none.
# The seed of a case is <name>.meta[0] of the fixture (make_golden advances it until no ReLU input lies within 1e-5 of zero, so
# that fp32 rounding cannot flip a ReLU switch between two correct implementations).
CASES = {
    "dc_4_8": ([(8, 4, 3), (8, 8, 3)], (1, 4, 64, 64)),
    "unet_down1": ([(64, 4, 3), (64, 64, 3)], (1, 4, 64, 64)),
    "unet_down2": ([(128, 64, 3), (128, 128, 3)], (1, 64, 32, 32)),
    "unet_up1": ([(64, 192, 3), (64, 64, 3)], (1, 192, 64, 64)),
    "unet_last": ([(1, 64, 1)], (1, 64, 64, 64)),
}


def _fill(rng, shape, bound):
    return ((rng.random(shape) * 2.0 - 1.0) * bound).astype(np.float32)


def case_tensors(name):
    """-> (params [(w, b)], x, r): r is dL/d(output)"""
    convs, xshape = CASES[name]
    seed = int(load()[f"{name}.meta"][0])
    rng = np.random.default_rng(seed)
    params = [(torch.from_numpy(_fill(rng, (co, ci, k, k), 1.0 / np.sqrt(ci * k * k))), torch.from_numpy(_fill(rng, (co,), 0.1)))
              for co, ci, k in convs]
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(_fill(rng, xshape, 1.5))
    yshape = (xshape[0], convs[-1][0]) + tuple(xshape[2:])
    r = torch.from_numpy(_fill(rng, yshape, 1.0))
    return params, x, r


def load():
    return np.load(GOLD)


def golden_err(z, name, key, got):
    """max |got - golden| / max(max|golden|, 1e-3) over what the fixture stores for <name>.<key> (full tensor, or lattice crop
    + fp64 checksums over the thinned axes)"""
    got = got.detach().double().cpu().numpy()
    errs = []
    pre = f"{name}.{key}."

    def rel(a, b):
        return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-3))
    if pre + "full" in z.files:
        errs.append(rel(got, z[pre + "full"].astype(np.float64)))
    if pre + "crop3" in z.files:
        errs.append(rel(got[..., ::3, ::3], z[pre + "crop3"].astype(np.float64)))
        errs.append(rel(got.sum(axis=(-2, -1)), z[pre + "chsum"]))
    if pre + "crop2" in z.files:
        errs.append(rel(got[::2, ::2], z[pre + "crop2"].astype(np.float64)))
        errs.append(rel(got.sum(axis=1), z[pre + "cisum"]))
    assert errs, f"no golden entry for {pre}*"
    return max(errs)


def emit_cube(seed, rows, cols, centers):
    """the cube of golden G10 (tests/golden/make_golden.py::g10_emit_cube), regenerated from the same PCG64 uniform streams"""
    rng = np.random.default_rng(seed)
    S = centers.size
    base = 1.0 + 5.0 * rng.random(S)
    raw = (base * (1.0 + 0.1 * (rng.random((rows, cols, S)) - 0.5))).astype(np.float32)
    dip = np.exp(-0.5 * ((centers - 2300.0) / 60.0) ** 2)
    raw[20:40, 3:6, :] *= (1.0 - 0.03 * dip).astype(np.float32)
    raw[:7, :3, :] = -9999.0
    raw[50, 7, 250] = -9999.0
    return raw


FULL_SEED = 4242
FULL_KEYS = [f"{blk}.{i}.{wb}" for blk in ("dconv_down1", "dconv_down2", "dconv_down3", "dconv_down4", "dconv_up3", "dconv_up2", "dconv_up1")
             for i in (0, 2) for wb in ("weight", "bias")] + ["conv_last.weight", "conv_last.bias"]
FULL_CH = {"dconv_down1": (4, 64), "dconv_down2": (64, 128), "dconv_down3": (128, 256), "dconv_down4": (256, 512),
           "dconv_up3": (768, 256), "dconv_up2": (384, 128), "dconv_up1": (192, 64)}


def full_unet_state():
    """state_dict of the reference's UNet(4, 1) as make_golden filled it: one PCG64 stream in state_dict order"""
    rng = np.random.default_rng(FULL_SEED)
    sd = {}
    for k in FULL_KEYS:
        blk, *rest = k.split(".")
        if blk == "conv_last":
            shape = (1, 64, 1, 1) if rest[0] == "weight" else (1,)
        else:
            ci, co = FULL_CH[blk]
            cin = ci if rest[0] == "0" else co
            shape = (co, cin, 3, 3) if rest[1] == "weight" else (co,)
        bound = np.sqrt(6.0 / (shape[1] * shape[2] * shape[3])) if len(shape) == 4 else 0.1
        sd[k] = torch.from_numpy(_fill(rng, shape, bound))
    return sd


def full_unet_input(tag):
    shape = tuple(int(v) for v in load()[f"unet_full.{tag}.shape"])
    return torch.from_numpy(_fill(np.random.default_rng(FULL_SEED + 1 + len(tag) + shape[0]), shape, 1.5))


def full_unet_grad_case():
    """input and upstream gradient of the golden backward case (unet_full.grad.*): L = sum(y * r); seed and shape from the fixture
    (make_golden advances the input seed until no ReLU / max-pool switch of the whole network sits within 1e-5 of flipping)"""
    meta = load()["unet_full.grad.meta"]
    seed, shape = int(meta[0]), tuple(int(v) for v in meta[1:])
    x = torch.from_numpy(_fill(np.random.default_rng(seed), shape, 1.5))
    r = torch.from_numpy(_fill(np.random.default_rng(seed + 77), (shape[0], 1) + shape[2:], 1.0))
    return x, r


def full_unet_grad_errs(z, grads):
    """{name: rel err} of parameter gradients (dict name -> tensor) against the golden marginal sums / bias gradients"""
    errs = {}
    for k in FULL_KEYS:
        g = grads[k].detach().double().cpu().numpy()
        if g.ndim == 1:
            want = z[f"unet_full.grad.{k}"]
            errs[k] = float(np.abs(g - want).max() / max(float(np.abs(want).max()), 1e-3))
        else:
            for tag, ax in (("cisum", 1), ("cosum", 0)):
                want = z[f"unet_full.grad.{k}.{tag}"]
                errs[f"{k}.{tag}"] = float(np.abs(g.sum(axis=ax) - want).max() / max(float(np.abs(want).max()), 1e-3))
    return errs
