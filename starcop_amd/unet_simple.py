"""The reference's in-repo plain U-Net (``starcop/models/architectures/unet.py:7-51`` ``UNet(n_channels, n_class)``, built from
``layer_factory.double_conv`` :4-9) on the HIP kernels -- forward and backward.  SURVEY.md section 8 row a18: the architecture is
not reachable from ``scripts/train.py`` (its branch in ``configure_architecture`` is commented out, model_module.py:226-236), but
it is the only network whose arithmetic the reference repository itself holds, so it doubles as a whole-network pin of the
convolution kernels (golden G9 ``unet_full.*``: the reference's own forward AND backward of the 7.78 M-parameter network).

    4 x [conv3x3(bias) + ReLU] x 2 encoder stages with MaxPool2d(2), three decoder stages
    [bilinear x2 (align_corners=True) -> cat(skip) -> double_conv], 1x1 head with bias; no BatchNorm.

``state_dict`` keys and shapes equal the reference module's (``dconv_down1.0.weight`` ... ``conv_last.bias``), so its checkpoints
load.  Execution follows the package's "normalise on load" model: every convolution stores its raw output; bias + ReLU are the
consumer's prologue (``SC_SRC_AFFINE`` with scale 1, shift = bias) and their backward is ``SC_SRC_BNBWD`` with (A, B, D) = (1, 0, 0);
a bias gradient is the first column of ``sc_bn_bwd_reduce``'s sums; MaxPool and the bilinear upsampling are ``sc_maxpool2x2`` /
``sc_upsample_bilinear2x`` and their ``_bwd`` twins; the skip concat is two sources of one convolution and, backwards, a channel
split of the data gradient.  3x3 layers run on the split 16-bit-MFMA kernels with three bf16 terms (fp32's exponent range:
activations are not normalised in this network); the 4-channel first layer's weight gradient and the 1x1 head use the fp32 MFMA
kernels.  There is no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, SC_CST, SRC_AFFINE, SRC_BNBWD, SRC_RAW, STAT_BNBWD, check, make_src, ptr, sc_conv_args,
                   sc_wgrad_args, stream)

TERMS = 3          # three bf16 terms per operand in the split kernels: no range assumptions on un-normalised activations


def _double_conv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(cout, cout, 3, padding=1), nn.ReLU(inplace=True))


def _cot(m, ks):
    return 64 if (m >= 64 or (ks == 1 and m > 32)) else 32


class SimpleUNet(nn.Module):
    def __init__(self, n_channels, n_class):
        super().__init__()
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

    # ------------------------------------------------------------------------------------------ packed filters
    def _pack(self, conv):
        """forward + backward-data packs and the bias constants of one convolution (channel counts padded to multiples of 8 with
        zero filters: the dense kernels read sources 8 channels at a time), rebuilt when the parameters change"""
        lib = _lib.load()
        key = id(conv)
        ver = (conv.weight._version, conv.weight.data_ptr(), conv.bias._version, conv.bias.data_ptr())
        ent = self._packs.get(key)
        if ent is not None and ent["ver"] == ver:
            return ent
        dev = conv.weight.device
        w = conv.weight.detach().float()
        co0, ci0, ks = w.shape[0], w.shape[1], w.shape[2]
        co, ci = -(-co0 // 8) * 8, -(-ci0 // 8) * 8
        wp = torch.zeros(co, ci, ks, ks, device=dev)
        wp[:co0, :ci0] = w
        cst = torch.zeros(co, SC_CST, device=dev)
        cst[:, 0] = 1.0
        cst[:co0, 1] = conv.bias.detach().float()
        cst[:, 2] = 1.0                                      # backward prologue: dy = 1 * [pass ? g : 0] + 0 * y + 0
        st = stream()
        ent = dict(ver=ver, w=wp, cst=cst, co=co, ci=ci, co0=co0, ci0=ci0, ks=ks)
        for tag, tflip, m in (("f", 0, co), ("b", 1, ci)):
            bx3 = ks == 3 and m >= 32
            cot = _cot(m, ks)
            if bx3:
                buf = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, cot, tflip, TERMS), device=dev)
                check(lib.sc_pack_weights_bx3(ptr(wp), ptr(buf), co, ci, cot, tflip, TERMS, st))
            else:
                if ks == 3 and m <= 16:
                    cot = 16
                buf = torch.empty(lib.sc_packed_weight_floats(co, ci, ks, cot, tflip), device=dev)
                check(lib.sc_pack_weights(ptr(wp), ptr(buf), co, ci, ks, cot, tflip, st))
            ent[tag] = (buf, cot, bx3)
        self._packs[key] = ent
        return ent

    # ------------------------------------------------------------------------------------------ kernel launches
    @staticmethod
    def _conv(srcs, pack, N, H, W, cout, ks, outs, csplit=None, accum=(0, 0)):
        lib = _lib.load()
        buf, cot, bx3 = pack
        a = sc_conv_args()
        a.nsrc = len(srcs)
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.wpk = buf.data_ptr()
        a.N, a.H, a.W, a.Cout, a.ks, a.co_t = N, H, W, cout, ks, cot
        a.out0 = outs[0].data_ptr()
        a.out1 = outs[1].data_ptr() if len(outs) > 1 else None
        a.csplit = cout if csplit is None else csplit
        a.accum0, a.accum1 = accum
        a.add0 = None; a.add1 = None; a.stats = None; a.absmax = None
        a.terms, a.down0 = TERMS, 0
        check((lib.sc_conv3x3_bx3 if bx3 else lib.sc_conv2d_mfma)(C.byref(a), stream()))

    @staticmethod
    def _wgrad(dy, srcs, N, H, W, cout, cin, ks):
        lib = _lib.load()
        a = sc_wgrad_args()
        a.dy = dy
        a.nsrc = len(srcs)
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.N, a.H, a.W, a.Cout, a.Cin, a.ks = N, H, W, cout, cin, ks
        bx3 = ks == 3 and cout >= 32 and cin >= 32
        n = lib.sc_wgrad_bx3_workspace_floats(N, H, W, cout, cin) if bx3 else lib.sc_wgrad_workspace_floats(N, H, W, cout, cin, ks)
        ws = torch.empty(n, device=torch.device("cuda", torch.cuda.current_device()))
        dw = torch.empty(cout, cin, ks, ks, device=ws.device)
        a.part, a.part_floats, a.dw = ws.data_ptr(), n, dw.data_ptr()
        a.terms, a.absmax = TERMS, None
        check((lib.sc_conv3x3_wgrad_bx3 if bx3 else lib.sc_conv2d_wgrad_mfma)(C.byref(a), stream()))
        return dw, ws

    # ------------------------------------------------------------------------------------------ forward
    def _run(self, x, save):
        _lib.require_device(x)
        lib = _lib.load()
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError(f"SimpleUNet: expected (N,{self.n_channels},H,W) input, got {tuple(x.shape)}")
        N, Cn, H, W = x.shape
        if H % 8 or W % 8:
            raise RuntimeError(f"SimpleUNet: H and W must be multiples of 8 (three MaxPool2d(2) stages), got {H}x{W}")
        st = stream()
        f32 = dict(dtype=torch.float32, device=x.device)
        keep = []                                        # tensors referenced through raw pointers stay alive until the sync
        pad = (-Cn) % 8
        x = x.contiguous().float()
        if pad:
            x = torch.cat([x, torch.zeros(N, pad, H, W, device=x.device)], 1).contiguous()
        rec = {}                                         # per conv: raw output y, input sources, geometry

        def conv(name, convm, srcs, h, w, act=ACT_RELU):
            ent = self._pack(convm)
            y = torch.empty((N, ent["co"], h, w), **f32)
            self._conv(srcs, ent["f"], N, h, w, ent["co"], ent["ks"], [y])
            rec[name] = dict(y=y, srcs=srcs, h=h, w=w, ent=ent, act=act)
            keep.append(y)
            return make_src(y, ent["co"], SRC_AFFINE, act=act, cst=ent["cst"])

        def block(name, blk, srcs, h, w):
            s = conv(name + ".0", blk[0], srcs, h, w)
            return conv(name + ".2", blk[2], [s], h, w)

        def pool(src, Cc, h, w):
            o = torch.empty((N, Cc, h // 2, w // 2), **f32)
            check(lib.sc_maxpool2x2(C.byref(src), ptr(o), N, Cc, h // 2, w // 2, st))
            keep.append(o)
            return o

        def up(src, Cc, h, w):
            o = torch.empty((N, Cc, 2 * h, 2 * w), **f32)
            check(lib.sc_upsample_bilinear2x(C.byref(src), ptr(o), N, Cc, h, w, st))
            keep.append(o)
            return o
        keep.append(x)
        c1 = block("dconv_down1", self.dconv_down1, [make_src(x, Cn + pad, SRC_RAW)], H, W)
        p1 = pool(c1, 64, H, W)
        c2 = block("dconv_down2", self.dconv_down2, [make_src(p1, 64, SRC_RAW)], H // 2, W // 2)
        p2 = pool(c2, 128, H // 2, W // 2)
        c3 = block("dconv_down3", self.dconv_down3, [make_src(p2, 128, SRC_RAW)], H // 4, W // 4)
        p3 = pool(c3, 256, H // 4, W // 4)
        c4 = block("dconv_down4", self.dconv_down4, [make_src(p3, 256, SRC_RAW)], H // 8, W // 8)
        u3 = up(c4, 512, H // 8, W // 8)
        d3 = block("dconv_up3", self.dconv_up3, [make_src(u3, 512, SRC_RAW), c3], H // 4, W // 4)
        u2 = up(d3, 256, H // 4, W // 4)
        d2 = block("dconv_up2", self.dconv_up2, [make_src(u2, 256, SRC_RAW), c2], H // 2, W // 2)
        u1 = up(d2, 128, H // 2, W // 2)
        d1 = block("dconv_up1", self.dconv_up1, [make_src(u1, 128, SRC_RAW), c1], H, W)
        head = conv("conv_last", self.conv_last, [d1], H, W, act=ACT_NONE)
        out = torch.empty((N, rec["conv_last"]["ent"]["co"], H, W), **f32)
        check(lib.sc_apply_src(C.byref(head), ptr(out), N, out.shape[1], H * W, st))
        logits = out[:, :self.n_class].contiguous()
        if save:
            return logits, dict(rec=rec, keep=keep, N=N, H=H, W=W, srcs=dict(c1=c1, c2=c2, c3=c3, c4=c4, d3=d3, d2=d2, d1=d1))
        torch.cuda.current_stream().synchronize()        # the raw-pointer tensors in `keep` go out of scope here
        return logits

    def forward(self, x):
        need_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if need_grad:
            return _SimpleUNetFunction.apply(self, x, *list(self.parameters()))
        with torch.no_grad():
            return self._run(x, save=False)

    # ------------------------------------------------------------------------------------------ backward
    def _backward(self, saved, g_out):
        """dL/dlogits -> {parameter name: gradient}; the autograd of architectures/unet.py:23-51 layer by layer"""
        lib = _lib.load()
        rec, N, H, W, S = saved["rec"], saved["N"], saved["H"], saved["W"], saved["srcs"]
        st = stream()
        dev = g_out.device
        f32 = dict(dtype=torch.float32, device=dev)
        keep = saved["keep"]
        grads = {}

        def conv_bwd(name, g, outs=None, csplit=None, accum=(0, 0)):
            """g = dL/d act(y + b) of conv `name`: parameter gradients, and the data gradient into outs if given"""
            r = rec[name]
            ent, y, h, w = r["ent"], r["y"], r["h"], r["w"]
            co, ci, ks = ent["co"], ent["ci"], ent["ks"]
            rows = lib.sc_stat_rows(STAT_BNBWD, N, h, w)
            sums = torch.zeros(rows, co, 2, dtype=torch.float64, device=dev)
            check(lib.sc_bn_bwd_reduce(ptr(g), ptr(y), ptr(ent["cst"]), r["act"], ptr(sums), N, co, h * w, None, None, st))
            grads[name + ".bias"] = sums.sum(0)[:ent["co0"], 0].float()
            dy = make_src(g, co, SRC_BNBWD, act=r["act"], cst=ent["cst"], aux=y)
            dw, ws = self._wgrad(dy, r["srcs"], N, h, w, co, ci, ks)
            grads[name + ".weight"] = dw[:ent["co0"], :ent["ci0"]]
            keep.extend([g, sums, dw, ws])
            if outs is not None:
                self._conv([dy], ent["b"], N, h, w, ci, ks, outs, csplit=csplit, accum=accum)
                keep.extend(outs)

        def block_bwd(name, g, h, w, outs=None, csplit=None, accum=(0, 0)):
            mid = rec[name + ".0"]["ent"]["co"]
            g_mid = torch.empty((N, mid, h, w), **f32)
            conv_bwd(name + ".2", g, [g_mid])
            conv_bwd(name + ".0", g_mid, outs, csplit, accum)

        def up_bwd(g_up, Cc, h, w):
            o = torch.empty((N, Cc, h, w), **f32)
            check(lib.sc_upsample_bilinear2x_bwd(ptr(g_up), ptr(o), N, Cc, h, w, st))
            keep.append(o)
            return o

        def pool_bwd(src, g_pool, g_in, Cc, h, w):          # g_in (full resolution h x w) += routed g_pool
            check(lib.sc_maxpool2x2_bwd(C.byref(src), ptr(g_pool), ptr(g_in), 1, N, Cc, h // 2, w // 2, st))
        co8 = rec["conv_last"]["ent"]["co"]
        g = torch.zeros((N, co8, H, W), **f32)
        g[:, :self.n_class] = g_out.float()
        g_d1 = torch.empty((N, 64, H, W), **f32)
        conv_bwd("conv_last", g, [g_d1])
        g_u1, g_c1 = torch.empty((N, 128, H, W), **f32), torch.empty((N, 64, H, W), **f32)
        block_bwd("dconv_up1", g_d1, H, W, [g_u1, g_c1], csplit=128)
        g_d2 = up_bwd(g_u1, 128, H // 2, W // 2)
        g_u2, g_c2 = torch.empty((N, 256, H // 2, W // 2), **f32), torch.empty((N, 128, H // 2, W // 2), **f32)
        block_bwd("dconv_up2", g_d2, H // 2, W // 2, [g_u2, g_c2], csplit=256)
        g_d3 = up_bwd(g_u2, 256, H // 4, W // 4)
        g_u3, g_c3 = torch.empty((N, 512, H // 4, W // 4), **f32), torch.empty((N, 256, H // 4, W // 4), **f32)
        block_bwd("dconv_up3", g_d3, H // 4, W // 4, [g_u3, g_c3], csplit=512)
        g_c4 = up_bwd(g_u3, 512, H // 8, W // 8)
        g_p3 = torch.empty((N, 256, H // 8, W // 8), **f32)
        block_bwd("dconv_down4", g_c4, H // 8, W // 8, [g_p3])
        pool_bwd(S["c3"], g_p3, g_c3, 256, H // 4, W // 4)
        g_p2 = torch.empty((N, 128, H // 4, W // 4), **f32)
        block_bwd("dconv_down3", g_c3, H // 4, W // 4, [g_p2])
        pool_bwd(S["c2"], g_p2, g_c2, 128, H // 2, W // 2)
        g_p1 = torch.empty((N, 64, H // 2, W // 2), **f32)
        block_bwd("dconv_down2", g_c2, H // 2, W // 2, [g_p1])
        pool_bwd(S["c1"], g_p1, g_c1, 64, H, W)
        block_bwd("dconv_down1", g_c1, H, W, None)
        keep.extend([g_p1, g_p2, g_p3])
        torch.cuda.current_stream().synchronize()
        return grads


class _SimpleUNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        logits, saved = net._run(x, save=True)
        ctx.net, ctx.saved = net, saved
        return logits

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        grads = net._backward(ctx.saved, g.contiguous())
        ctx.saved = None
        out = [grads[name].reshape(p.shape).contiguous() if p.requires_grad else None for name, p in net.named_parameters()]
        return (None, None) + tuple(out)
