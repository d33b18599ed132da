"""The reference's in-repo plain U-Net (``starcop/models/architectures/unet.py:7-51`` ``UNet(n_channels, n_class)``, built from
``layer_factory.double_conv`` :4-9) on the HIP kernels -- inference.  SURVEY.md section 8 row a18: the architecture is not
reachable from ``scripts/train.py`` (its branch in ``configure_architecture`` is commented out, model_module.py:226-236), but it
is the only network whose arithmetic the reference repository itself holds, so it doubles as a whole-network pin of the
convolution kernels (golden G9 ``unet_full.*``: the reference's own forward of the 7.78 M-parameter network).

    4 x [conv3x3(bias) + ReLU] x 2 encoder stages with MaxPool2d(2), three decoder stages
    [bilinear x2 (align_corners=True) -> cat(skip) -> double_conv], 1x1 head with bias; no BatchNorm.

``state_dict`` keys and shapes equal the reference module's (``dconv_down1.0.weight`` ... ``conv_last.bias``), so its checkpoints
load.  Execution follows the package's "normalise on load" model: every convolution stores its raw output; bias + ReLU are the
consumer's prologue; MaxPool and the bilinear upsampling are ``sc_maxpool2x2`` / ``sc_upsample_bilinear2x``; the skip concat is
two sources of one convolution.  3x3 layers run on the split 16-bit-MFMA kernels with three bf16 terms (fp32's exponent range:
activations are not normalised in this network), the 1x1 head on the fp32 MFMA.  There is no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, SC_CST, SRC_AFFINE, SRC_RAW, check, make_src, ptr, sc_conv_args, stream


def _double_conv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(cout, cout, 3, padding=1), nn.ReLU(inplace=True))


class SimpleUNet(nn.Module):
    def __init__(self, n_channels, n_class):
        super().__init__()
        if n_class != 1 and n_class % 8:
            raise ValueError("SimpleUNet: n_class must be 1 or a multiple of 8 on the HIP head")
        self.n_channels, self.n_class = n_channels, n_class
        self.dconv_down1 = _double_conv(n_channels, 64)
        self.dconv_down2 = _double_conv(64, 128)
        self.dconv_down3 = _double_conv(128, 256)
        self.dconv_down4 = _double_conv(256, 512)
        self.dconv_up3 = _double_conv(256 + 512, 256)
        self.dconv_up2 = _double_conv(128 + 256, 128)
        self.dconv_up1 = _double_conv(128 + 64, 64)
        self.conv_last = nn.Conv2d(64, n_class, 1)
        self._packs = {}

    # ---- packed filters, rebuilt when a parameter changes
    def _pack(self, conv, cin_pad=0):
        lib = _lib.load()
        key = id(conv)
        ver = (conv.weight._version, conv.weight.data_ptr(), conv.bias._version)
        ent = self._packs.get(key)
        if ent is not None and ent["ver"] == ver:
            return ent
        w = conv.weight.detach().float().contiguous()
        if cin_pad:
            w = torch.cat([w, torch.zeros(w.shape[0], cin_pad, *w.shape[2:], device=w.device)], 1).contiguous()
        co, ci, ks = w.shape[0], w.shape[1], w.shape[2]
        cst = torch.zeros(co, SC_CST, device=w.device)
        cst[:, 0], cst[:, 1] = 1.0, conv.bias.detach().float()
        if ks == 3:
            co_t = 64 if co >= 64 else 32
            wpk = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, co_t, 0, 3), device=w.device)
            check(lib.sc_pack_weights_bx3(ptr(w), ptr(wpk), co, ci, co_t, 0, 3, stream()))
        else:
            co_t = 32 if co <= 32 else 64
            wpk = torch.empty(lib.sc_packed_weight_floats(co, ci, 1, co_t, 0), device=w.device)
            check(lib.sc_pack_weights(ptr(w), ptr(wpk), co, ci, 1, co_t, 0, stream()))
        ent = dict(ver=ver, wpk=wpk, co_t=co_t, cst=cst, co=co, ks=ks, w=w)
        self._packs[key] = ent
        return ent

    def _conv(self, srcs, conv, N, H, W, cin_pad=0):
        """raw convolution output and the source (bias + ReLU prologue) its consumers read it through"""
        lib = _lib.load()
        ent = self._pack(conv, cin_pad)
        y = torch.empty((N, ent["co"], H, W), dtype=torch.float32, device=ent["wpk"].device)
        a = sc_conv_args()
        a.nsrc = len(srcs)
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.wpk = ent["wpk"].data_ptr()
        a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, ent["co"], ent["ks"], ent["co_t"]
        a.out0 = y.data_ptr(); a.out1 = None
        a.csplit, a.accum0, a.accum1 = ent["co"], 0, 0
        a.add0 = None; a.add1 = None; a.stats = None
        a.terms = 3
        check((lib.sc_conv3x3_bx3 if ent["ks"] == 3 else lib.sc_conv2d_mfma)(C.byref(a), stream()))
        self._keep += [y, ent["cst"]]
        return y, make_src(y, ent["co"], SRC_AFFINE, act=ACT_RELU, cst=ent["cst"])

    def _block(self, srcs, block, N, H, W, cin_pad=0):
        _, s = self._conv(srcs, block[0], N, H, W, cin_pad)
        return self._conv([s], block[2], N, H, W)

    @torch.no_grad()
    def forward(self, x):
        _lib.require_device(x)
        lib = _lib.load()
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("SimpleUNet runs inference on the HIP kernels (the reference never trains this architecture: "
                                      "model_module.py:226-236); call .eval() / torch.no_grad()")
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError(f"SimpleUNet: expected (N,{self.n_channels},H,W) input, got {tuple(x.shape)}")
        N, Cn, H, W = x.shape
        if H % 8 or W % 8:
            raise RuntimeError(f"SimpleUNet: H and W must be multiples of 8 (three MaxPool2d(2) stages), got {H}x{W}")
        self._keep = []
        pad = (-Cn) % 8                                   # the dense kernels read sources in multiples of 8 channels
        x = x.contiguous().float()
        if pad:
            x = torch.cat([x, torch.zeros(N, pad, H, W, device=x.device)], 1).contiguous()
        st = stream()
        f32 = dict(dtype=torch.float32, device=x.device)

        def pool(src, Cc, h, w):
            o = torch.empty((N, Cc, h // 2, w // 2), **f32)
            check(lib.sc_maxpool2x2(C.byref(src), ptr(o), N, Cc, h // 2, w // 2, st))
            return o

        def up(src, Cc, h, w):
            o = torch.empty((N, Cc, 2 * h, 2 * w), **f32)
            check(lib.sc_upsample_bilinear2x(C.byref(src), ptr(o), N, Cc, h, w, st))
            return o
        _, c1 = self._block([make_src(x, Cn + pad, SRC_RAW)], self.dconv_down1, N, H, W, cin_pad=pad)
        p1 = pool(c1, 64, H, W)
        _, c2 = self._block([make_src(p1, 64, SRC_RAW)], self.dconv_down2, N, H // 2, W // 2)
        p2 = pool(c2, 128, H // 2, W // 2)
        _, c3 = self._block([make_src(p2, 128, SRC_RAW)], self.dconv_down3, N, H // 4, W // 4)
        p3 = pool(c3, 256, H // 4, W // 4)
        _, c4 = self._block([make_src(p3, 256, SRC_RAW)], self.dconv_down4, N, H // 8, W // 8)
        u3 = up(c4, 512, H // 8, W // 8)
        _, d3 = self._block([make_src(u3, 512, SRC_RAW), c3], self.dconv_up3, N, H // 4, W // 4)
        u2 = up(d3, 256, H // 4, W // 4)
        _, d2 = self._block([make_src(u2, 256, SRC_RAW), c2], self.dconv_up2, N, H // 2, W // 2)
        u1 = up(d2, 128, H // 2, W // 2)
        _, d1 = self._block([make_src(u1, 128, SRC_RAW), c1], self.dconv_up1, N, H, W)
        # 1x1 head: Cout = n_class is padded to 8 output channels for the dense kernel, bias added on the way out
        head = self.conv_last
        ent = self._packs.get(("head", head.weight._version, head.weight.data_ptr(), head.bias._version))
        if ent is None:
            co8 = -(-self.n_class // 8) * 8
            w = torch.zeros(co8, 64, 1, 1, device=x.device); w[:self.n_class] = head.weight.detach().float()
            wpk = torch.empty(lib.sc_packed_weight_floats(co8, 64, 1, 32, 0), device=x.device)
            check(lib.sc_pack_weights(ptr(w), ptr(wpk), co8, 64, 1, 32, 0, st))
            cst = torch.zeros(co8, SC_CST, device=x.device); cst[:, 0] = 1.0; cst[:self.n_class, 1] = head.bias.detach().float()
            ent = dict(wpk=wpk, cst=cst, co=co8, w=w)
            self._packs = {k: v for k, v in self._packs.items() if not (isinstance(k, tuple) and k[0] == "head")}
            self._packs[("head", head.weight._version, head.weight.data_ptr(), head.bias._version)] = ent
        y = torch.empty((N, ent["co"], H, W), **f32)
        a = sc_conv_args()
        a.nsrc = 1; a.src[0] = d1
        a.wpk = ent["wpk"].data_ptr()
        a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, ent["co"], 1, 32
        a.out0 = y.data_ptr(); a.out1 = None
        a.csplit, a.accum0, a.accum1 = ent["co"], 0, 0
        a.add0 = None; a.add1 = None; a.stats = None
        check(lib.sc_conv2d_mfma(C.byref(a), st))
        out = torch.empty_like(y)
        s = make_src(y, ent["co"], SRC_AFFINE, act=ACT_NONE, cst=ent["cst"])
        check(lib.sc_apply_src(C.byref(s), ptr(out), N, ent["co"], H * W, st))
        self._keep = []
        return out[:, :self.n_class].contiguous()
