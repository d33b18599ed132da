"""HyperSTARCOP network: the MobileNetV2-encoder U-Net of ``smp.Unet('mobilenet_v2')`` executed
by hand-written HIP kernels (libstarcop_hip.so) -- forward, backward, eval and train mode.

Reference: ``configure_architecture`` builds ``smp.Unet(encoder_name='mobilenet_v2',
encoder_weights=None, in_channels=C, classes=1, activation=None)``
(/root/reference/starcop/models/model_module.py:224-256).  This module keeps

  * the ``state_dict`` key names/shapes of that network (``encoder.features.N...``,
    ``decoder.blocks.N.convK.M``, ``segmentation_head.0``) so reference checkpoints load,
  * ordinary ``nn.Parameter`` tensors so ``torch.optim.Adam(network.parameters())`` works
    (model_module.py:174), ``.train()/.eval()`` BatchNorm semantics, ``.to(device)``.

The ``torch.nn`` sub-modules are parameter containers only: no torch op computes anything on the
path.  Execution model ("normalise on load"): every convolution stores its RAW output and
accumulates per-channel sum / sum-of-squares in its epilogue; BatchNorm + ReLU/ReLU6 of a producer
are applied by the consumer while it stages tiles into LDS, nearest x2 upsampling and the skip
concat are address arithmetic in the consumer, and the backward pass applies the BatchNorm /
activation backward the same way (g, y -> dy on load).
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_RELU6, SC_CST, sc_irb_args, sc_irt_args, SRC_AFFINE, SRC_BNBWD, SRC_NORM, SRC_RAW, STAT_BNBWD, sc_wgrad_pending,
                   PACK_PW3, PACK_SP, PACK_SPD, PACK_THIN16, STAT_PW3, STAT_CONV1, STAT_CONV1K, STAT_CONV3, STAT_DW, STAT_STEM, TERMS_F16X2, check, make_src, ptr, sc_bn_tail, sc_bnr_args,
                   sc_conv_args, sc_wgrad_args, stream)

MBV2_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                 (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
DECODER_CHANNELS = (256, 128, 64, 32, 16)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


# ----------------------------------------------------------------------------------------------
# parameter containers (names == smp / torchvision names)
def _conv_bn_act(cin, cout, k, stride=1, groups=1, act="relu6"):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOMENTUM),
                         nn.ReLU6() if act == "relu6" else nn.ReLU())


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = cin * t
        self.use_res_connect = stride == 1 and cin == cout
        self.stride, self.expand = stride, t != 1
        layers = []
        if t != 1:
            layers.append(_conv_bn_act(cin, hid, 1))
        layers += [_conv_bn_act(hid, hid, 3, stride, groups=hid),
                   nn.Conv2d(hid, cout, 1, bias=False),
                   nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOMENTUM)]
        self.conv = nn.Sequential(*layers)


class _Encoder(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        feats = [_conv_bn_act(in_channels, 32, 3, 2)]
        cin = 32
        for t, c, n, s in MBV2_SETTINGS:
            for i in range(n):
                feats.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_conv_bn_act(cin, 1280, 1))
        self.features = nn.Sequential(*feats)


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = _conv_bn_act(cin + cskip, cout, 3, act="relu")
        self.conv2 = _conv_bn_act(cout, cout, 3, act="relu")


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        cins = (1280,) + DECODER_CHANNELS[:-1]
        cskips = (96, 32, 24, 16, 0)
        self.blocks = nn.ModuleList(_DecoderBlock(i, s, o) for i, s, o in zip(cins, cskips, DECODER_CHANNELS))


# ----------------------------------------------------------------------------------------------
class _T:
    """An activation of the plan.  kind: 'input' | 'raw' (conv output, BN+act applied by readers) | 'fin'."""
    __slots__ = ("name", "C", "shift", "kind", "bn", "act", "buf", "grad", "cst", "cstb", "stats", "bsums",
                 "grad_written", "alias_grad_of", "res_grad")

    def __init__(self, name, Cn, shift, kind, bn=None, act=ACT_NONE):
        self.name, self.C, self.shift, self.kind, self.bn, self.act = name, Cn, shift, kind, bn, act
        self.buf = self.grad = self.cst = self.cstb = self.stats = self.bsums = None
        self.grad_written = False
        self.alias_grad_of = None   # raw-linear tensor whose grad is the grad of the residual sum
        self.res_grad = None        # 'fin' tensor z = self + ...: z.grad must be added to self.grad


_BN_PRE = True      # coalesced pre-reduction of many-row BatchNorm statistics (sc_bn_finalize scratch)
_COT_RATIO = 1.15      # 64-wide cout tiles when they pad the channel count by less than this factor (measured per layer)
_COT_RATIO3 = 1.15     # the same choice for the 3x3 layers
_HEAD_FUSED_BWD = True     # gin, dW and dbias of the head in one sweep (False: the separate data / weight gradient launches)


def _pick_cot(M, ks=1):
    if M <= 16 and ks == 3:
        return 16            # thin layers: v_mfma_f32_16x16x4_f32 kernel, no wasted MFMA rows
    if M <= 32:
        return 32
    p32, p64 = -(-M // 32) * 32, -(-M // 64) * 64
    return 64 if p64 <= p32 * (_COT_RATIO3 if ks == 3 else _COT_RATIO) else 32


# Pointwise convolutions on sc_conv1x1_pw3 / sc_conv1x1_wgrad_pw3 (conv_pw3.hip: split-bf16 MFMA, register-only, no LDS staging)
# or on k_conv_mfma<1> / k_conv1_ksplit / k_wgrad_mfma<1> (fp32 MFMA, LDS staging, K split over the waves of a work-group).
# Measured per layer and pass at batch 4, 16 and 64 (tools/bench_layers.py with STARCOP_PW3=all against =0: profiles/r03a/b_layers_*,
# profiles/r04_layers_b{4,64}_pw3_{all,0}.txt).  What decides is the plane (H*W), the contraction length K and -- new in round 4 --
# how many waves the launch has: the register-only family does not split K, so a long contraction only pays once
# (32-pixel blocks of the batch) x (32-channel output blocks) fills the machine:
#   forward        planes up to 64^2; K <= 320 always, longer K from 1024 (pixel block, output block) pairs
#                  (batch 4: the 32^2 / 16^2 projections stay on the K-splitting kernel, 15-21 vs 23-43 us; batch 64: every one moves
#                  over, 72 vs 103 us on features.15-17)
#   backward-data  K <= 192 (the projections' gradients) on planes up to 64^2 while the batch has <= 131072 pixels of that plane size;
#                  longer K (the expansions' gradients) at <= 32^2 from 2048 pairs (batch 64: 48 vs 73, 83 vs 130, 148 vs 222 us;
#                  batch 16 and 4: the other family, 27 vs 32, 38 vs 68)
#   backward-weight  planes up to 32^2 of at most 32768 batch pixels, small filters (K*M bounds as measured at batch 16)
# "1" = these rules, "all" = every pointwise launch (the tests run both), "0" = off.
_PW3 = os.environ.get("STARCOP_PW3", "1")
# 64 x 64 planes with a short contraction on the streaming kernel of sc_conv2d_mfma (k_pw_stream) instead of sc_conv1x1_pw3: the
# features.5-.7 expansions forward 25-27 -> 21-22 us, the features.4-.6 projections' data gradients 23-30 -> 19-25 us at batch 16
# (tools/bench_layers.py, same box); step level to +0.6 % (four alternating pairs).  "0": the round-5 rule
_PWS64 = os.environ.get("STARCOP_PWS64", "1") == "1"


def _use_pw3(which, N, HW, Cin, Cout):
    """which: 0 forward, 1 backward-data, 2 backward-weight (reads 8-pixel groups: H*W % 8 == 0) of a Cin -> Cout pointwise layer on
    N planes of HW pixels"""
    if _PW3 == "0" or (which == 2 and HW % 8):
        return False
    if _PW3 == "all":
        return True
    NP = N * HW
    npb = -(-NP // 32)
    if _PWS64 and HW == 4096 and which in (0, 1):
        # 64 x 64 planes with a short contraction (the features.5-.7 expansions forward, the features.4-.6 projections' data gradients):
        # the streaming kernel inside sc_conv2d_mfma (k_pw_stream)
        K_, M_ = (Cin, Cout) if which == 0 else (Cout, Cin)
        if K_ in (16, 24, 32) and M_ <= 192 and (which == 1 or K_ in (24, 32)):
            return False
    if which == 0:
        return HW <= 4096 and (Cin <= 320 or npb * (-(-Cout // 32)) >= 1024)
    if which == 1:
        K, MB = Cout, -(-Cin // 32)
        return (K <= 192 and HW <= 4096 and NP <= 131072) or (K > 192 and HW <= 1024 and npb * MB >= 2048)
    KM = Cin * Cout
    return HW <= 1024 and NP <= 32768 and KM <= 300000 and (HW <= 256 or KM <= 32768)


# Fused TRAINING execution of the expansion + depthwise pair of a stride-2 inverted-residual block (conv_irt.hip: the 6x-expanded
# tensor and its gradient are never stored, every sweep recomputes it from the block input).  Measured per block (tools/bench_irt.py
# against the launches it replaces, tools/bench_layers.py) and on the whole step (bench.py with STARCOP_IRT=1 / 0, same box):
#   batch 16: features.2 (16 -> 96 channels at 256^2: e = 403 MB) 610 vs 835 us, step +1.3-1.5 %; features.4 (128^2, 151 MB) 414 vs
#             341, features.7 (64^2) 244 vs 143
#   batch 64: features.2 (1.6 GB) step +2.5 % (1555 vs 1517 tiles/s); features.4 (604 MB) 1292 vs 1278: a tie; features.7 loses
#   batch  4: features.2 (101 MB -- the tensor lives in the 256 MB Infinity Cache, the separate kernels' passes over it are cache
#             hits) step -4..6 % (641 vs 674 tiles/s)
# so the recomputation pays where the expanded tensor is large against the cache AND the plane is large: planes >= 256^2 whose e is
# >= 256 MB.  "1" = that rule, "all" = every supported block (tests), "0" = off.
_IRT = os.environ.get("STARCOP_IRT", "1")


def _use_irt(N, Cin, Hd, Hin, Win, stride):
    if _IRT == "0" or not _lib.load().sc_irt_supported(Cin, Hd, Hin, Win, stride):
        return False
    return _IRT == "all" or (Hin * Win >= 65536 and 4 * N * Hd * Hin * Win >= (256 << 20))


# INFERENCE: a stride-1 inverted-residual block (expansion -> depthwise -> projection [-> residual add]) as ONE launch (conv_irb.hip:
# the expanded tensors never leave the CU).  Eval-mode BatchNorm is a per-channel affine, so nothing batch-wide separates the three
# convolutions; at 32 x 32 / 16 x 16 planes the separate launches are 10-35 us each for 3-8 us of work (per block, batch 16, us,
# tools/bench_eval_layers.py, separate -> fused: see DESIGN 16).  "1" = the measured rule, "all" = every supported block (tests),
# "0" = off.
_IRB = os.environ.get("STARCOP_IRB", "1")


def _use_irb(N, Cin, hidden, Cout, H, W, stride):
    """The rule: blocks whose patch fits two work-groups per CU (Cin <= 96, hidden % 64 == 0, Cout <= 128: features.5 / .6 / .8-.13) on
    planes up to 80 x 80, and the stride-2 blocks features.7 / .14 (H, W: the INPUT plane).  features.15-.17 (Cin = 160 at 16 x 16) are NOT taken: every work-group streams the block's whole filter set
    (1.8 MB as three bf16 terms) for a 4 x 8 pixel tile, 128 work-groups of them -- 80-98 us against 74-86 us for the three launches
    (tools/bench_irb.py); splitting the hidden channels over work-groups instead of the pixels is what that shape needs."""
    if _IRB == "0" or not _lib.load().sc_irb_supported(Cin, hidden, Cout, H, W, stride):
        return False
    if _IRB == "all":
        return True
    if stride == 2:       # features.7 (32 -> 192 -> 64 from 64 x 64): 41 vs 54 us; features.14 (96 -> 576 -> 160 from 32 x 32): 74 vs 75 -- not taken
        return Cin <= 64 and hidden % 64 == 0 and H * W <= 25600
    return Cin <= 96 and hidden % 64 == 0 and Cout <= 128 and H * W <= 6400


# Decoder conv1 = conv3x3(cat([nearest_up2(prev), skip])) as a SUB-PIXEL convolution (conv_sp.hip: four phase-specific 2x2
# convolutions on the low-resolution prev, the skip channels as low-resolution parity planes: 2.25x fewer MFMAs for the up-sampled
# channels, every low-resolution value staged once instead of once per high-resolution copy).  Measured against sc_conv3x3_bx3 with
# an up-sampled source (tools/bench_sp.py, us per launch, batch 16): decoder.blocks.0 311 -> 288 (16 x 16 low-resolution planes: 128
# work-groups of 8 waves for 256 CUs; 1131 -> 778 at batch 64), blocks.1 125 -> 79, blocks.2 147 -> 104, blocks.3 178 -> 122; blocks.4
# (16 output channels: half-empty MFMA rows) 196-213, level with sc_conv3x3_thin16's 202, stays there.  "1" = that rule, "all" =
# every decoder conv1 (tests), "0" = off.
_SP = os.environ.get("STARCOP_SP", "1")
# BatchNorm-backward sums of a decoder tensor from the data-gradient launch that writes its gradient (sc_bnr_args) instead of the separate
# sc_bn_bwd_reduce pass over (gradient, y).  Built, parity-tested and measured in round 5 (tools/bench_layers.py, batch 16, us): the pass it
# removes streams at 5.3 TB/s, the extra read of y in a convolution epilogue runs at that kernel's 2.3-3 TB/s -- decoder.blocks.4.conv2
# data gradient 197 -> 302 for a 120 -> 10 BatchNorm pass, blocks.3.conv2 134 -> 193 for 60 -> 10, blocks.4.conv1 (2x2 down-summed store,
# every other lane idle) 244 -> 392 for 59 -> 13: the step loses 1 % (1444 / 1436 vs 1460 / 1454 tiles/s, same box, alternating).  OFF by
# default; "1" enables it (tests/test_gpu_unet.py runs the network both ways).
_BNR = os.environ.get("STARCOP_BNR", "0") == "1"
# ... for the thin layer alone (decoder.blocks.4.conv2's data gradient leaves conv1's sums; y requested ahead of the MFMA phase since round 6)
_BNR_THIN = os.environ.get("STARCOP_BNR_THIN", "0") == "1"
_THIN_SPD = os.environ.get("STARCOP_THIN_SPD", "1") == "1"      # (same-box A/B of decoder.blocks.4.conv1's sub-pixel data gradient)
# training steps pack the decoder's / the backward filter layouts on the weight-gradient stream, beside the encoder's forward ("0": on
# the main stream, ahead of the forward -- A/B)
_PACK_SIDE = os.environ.get("STARCOP_PACK_SIDE", "1") == "1"
# depthwise layers finalize their own BatchNorm in the launch's tail (sc_dwconv3x3_fwd_bn: last arrival per channel by ticket) when a
# channel has at most this many statistics rows (15 of the 17 depthwise layers at batch 16 have 16-32).  Built, parity-tested and
# measured in round 5: with a device-scope fence before the ticket the step went 1460 -> 1060 tiles/s (every arriving work-group
# writes back its XCD's L2); with write-through stores + a relaxed ticket it is correct and LEVEL (1470.0 / 1469.2 vs 1473.7 / 1472.4
# same-box): every work-group now waits ~2-3 us for its ticket to come back from memory, which is what the 15 removed ~6 us launches
# had cost.  0 = off (default): the separate sc_bn_finalize launch.
_DW_TAIL_ROWS = int(os.environ.get("STARCOP_DW_TAIL_ROWS", "0"))
_SP_SKIPTILES = os.environ.get("STARCOP_SP_SKIPTILES", "1") == "1"      # (same-box A/B of decoder.blocks.0's one-launch data gradient)


def _experiment(name, default=""):
    """Elimination experiments (STARCOP_EXP_*: launches skipped or bytes halved -- results WRONG, timing valid) are measuring tools, not
    product switches: a stray variable must not silently skip weight gradients.  They are honoured only together with
    STARCOP_EXPERIMENT_OK=1 (tools/ab_*.sh set it); without it the import fails, with it every active one is announced once."""
    v = os.environ.get(name, default)
    if v not in ("", "0"):
        if os.environ.get("STARCOP_EXPERIMENT_OK") != "1":
            raise RuntimeError(f"{name}={v} is an elimination experiment (wrong results by design); set STARCOP_EXPERIMENT_OK=1 to run it")
        import warnings
        warnings.warn(f"starcop_amd: elimination experiment {name}={v} is ACTIVE -- results of this process are wrong by design")
    return v


_EXP_NO_WGRAD = _experiment("STARCOP_EXP_NO_WGRAD", "0") == "1"
# elimination experiment (results WRONG after the first steps, timing valid): skip the BatchNorm finalize launches of the training forward
# ("f"), of the backward ("b") or both ("fb") from the fourth step of a plan on -- what the ~93 dependent ~5 us launches cost the step
_EXP_NO_BNFIN = _experiment("STARCOP_EXP_NO_BNFIN", "")
_EXP_SIDE2 = _experiment("STARCOP_EXP_SIDE2", "0") == "1"      # tools/: elimination experiment only


_SP_TERMS = (TERMS_F16X2, 1)      # arithmetic modes of the sub-pixel forward / data-gradient kernels (conv_sp.hip)


def _use_sp(N, Ho, Wo, Cout):
    """forward of a decoder conv1 with an Ho x Wo output on sc_conv3x3_sp?"""
    if _SP == "0" or Ho % 2 or Wo % 2:
        return False
    if _SP == "all":
        return True
    Hl, Wl = Ho // 2, Wo // 2
    tw = 32 if Wl >= 32 else 16
    wgs = N * (-(-Wl // tw)) * (-(-Hl // (256 // tw))) * (-(-Cout // 32))
    return Cout >= 32 and wgs >= 128


def _use_spd(N, Ho, Wo, Cup, Csk=0):
    """data gradient of a decoder conv1's up-sampled channels on sc_conv3x3_sp_dgrad?  (tools/bench_sp.py, us, batch 16, against
    sc_conv3x3_bx3(down0) on the same channels: decoder.blocks.0 336 -> 250, blocks.1 140 -> 129, blocks.2 148 -> 83; blocks.3 / .4 have
    64 / 32 such channels for the kernel's 128-channel tiles: 176 -> 190, 285 -> 500 -- they keep the 3x3 form)"""
    if _SP == "0" or Ho % 2 or Wo % 2:
        return False
    return _SP == "all" or Cup >= 128 or (Csk and _lib.load().sc_spd_vskip_ok(Cup, Csk))       # (.. or ONE launch for both gradients)


def _use_spd_skip_tiles(Cup, Csk):
    """the skip channels' gradient of a decoder conv1 as additional channel tiles of its sub-pixel data-gradient launch (instead of a
    3x3 launch of its own on 32-wide tiles, which stages dy again)?  Each skip tile costs what a tile of 128 up-sampled channels costs,
    so it pays where the skip channels are few tiles beside many: decoder.blocks.0 (10 + 3 tiles; the 3x3 launch it replaces: 54 us at
    batch 16).  decoder.blocks.1 (2 + 1): level; blocks.2 (1 + 1: twice the launch for a 50 us one): no."""
    if _SP == "all":
        return True
    return _SP != "0" and _SP_SKIPTILES and Cup >= 1024 and -(-Csk // 32) * 3 <= -(-Cup // 128)


def _use_spw(N, Ho, Wo, Cout, Cup):
    """weight gradient of a decoder conv1's up-sampled channels as the box-sum GEMM of conv_spw.hip?  up(x) is constant over 2x2 output
    blocks, so dW[kh][kw] = sum_q x[q] S_(kh,kw)[q] with S tap-aligned 2x2 box sums of dy: nine plain GEMMs over the low-resolution
    pixels, a quarter of the multiply-adds -- but the nine box-sum planes (2.25x the size of dy, as two fp16 terms) are written and
    read once, which only pays where the layer is matrix-bound (tools/bench_sp.py, us at batch 16, up-sampled channels alone, 3x3 form
    -> box-sum GEMM: decoder.blocks.0 338 -> 222, blocks.1 143 -> 147, blocks.2 148 -> 228, blocks.3 186 -> 482)"""
    if _SP == "0" or Ho % 2 or Wo % 2:
        return False
    return _SP == "all" or Cup >= 512


def _use_ksplit(N, HW, K, M, ks=1):
    """1x1 layers with few pixels and long K: the split-K kernel (sc_conv1x1_ksplit) beats the 128-pixel tiles when
    those leave most CUs idle (measured crossover, tools/bench_pw.py)."""
    if ks != 1:
        return False
    wgs = N * (-(-HW // 128)) * (-(-M // _pick_cot(M, 1)))
    return wgs <= 512 and (K >= 384 or (K >= 192 and wgs <= 256))


class HyperStarcopUNet(nn.Module):
    """Drop-in for ``ModelModule.network`` (smp.Unet mobilenet_v2), running on libstarcop_hip.so."""

    def __init__(self, in_channels=4, classes=1):
        super().__init__()
        if classes != 1:
            raise ValueError("HyperStarcopUNet: the HIP segmentation head implements classes=1 "
                             "(starcop/config.yaml:40 num_classes: 1)")
        if not (1 <= in_channels <= 8):
            raise ValueError("HyperStarcopUNet: in_channels must be in [1, 8]")
        self.in_channels = in_channels
        self.encoder = _Encoder(in_channels)
        self.decoder = _Decoder()
        self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        self.reset_parameters()
        self._ops, self._tensors = self._build_ops()
        self._plans = {}
        self._pflat = self._gflat = None
        self._pack_version = None
        self.register_load_state_dict_post_hook(HyperStarcopUNet._after_load)

    @staticmethod
    def _after_load(module, incompatible_keys):
        module.check_split_range()          # (a post hook must return None)

    # -- range of the default (two-fp16-term) split: filters are scaled by 2^8 and activations by 2 before the fp16 conversion
    FP16_MAX_WEIGHT, FP16_MAX_ACT = 255.0, 32752.0
    range_check_every = 200     # optimiser steps between two check_split_range() calls during training (0: never)

    def _split_feeders(self):
        """inputs of the split 3x3 convolutions: (BatchNorm'd tensors whose activation is not bounded by ReLU6, residual sums)"""
        raw, fin = [], []
        for op in self._ops:
            if op["type"] == "conv3":        # every decoder convolution stages fp16 terms (k_conv3_bx3 / _ws / k_conv3_thin_h)
                for t in op["ins"]:
                    if t.kind == "fin" and t.name not in fin:
                        fin.append(t.name)
                    elif t.kind == "raw" and t.act != ACT_RELU6 and t.name not in raw:
                        raw.append(t.name)
        return raw, fin

    def split_range_report(self, sync_ranks=False):
        """The operands of the split 3x3 convolutions against the limits of the two-fp16-term kernels:

        * largest |filter| (limit 255: filters are scaled by 2^8 before the fp16 conversion);
        * ``activation_observed``: the largest |activation| that has reached one of them so far, from DEVICE-SIDE STICKY RECORDS --
          for BatchNorm-fed inputs ``sc_bn_bwd_reduce / sc_bn_bwd_small`` leave max |BN(y)| of every training step (they stream y
          anyway), in inference ``sc_add_srcs_absmax(out=NULL)`` records it at the ``range_check_every`` cadence; residual sums
          (no BatchNorm bounds them) are recorded by ``sc_add_srcs_absmax`` in every forward.  The records are never lowered, so a
          check at any later time sees every step since the last one: there is no window in which a clamp goes unnoticed;
        * ``activation_bound``: the static bound max_c(64 |gamma_c| + |beta_c|) (|x_hat| <= 64 covers every realistic tile) --
          what can be said about a checkpoint before any data has flowed (``load_state_dict``).

        Synchronises the device.  ``sync_ranks``: MAX over the ranks of an initialised process group, so that every replica of a
        data-parallel job takes the same decision (FusedAdam passes it; all ranks step together there).  The all-reduce also
        carries ``switched`` -- 1 if ANY rank has already left the two-fp16-term mode on its own (an inference forward on
        rank-local data may do that, ``_forward_impl``) -- and is issued by every rank whatever its own precision, so the
        collective sequence of the replicas never diverges."""
        wmax = amax = rmax = omax = 0.0
        raw_feed, fin_feed = self._split_feeders()
        for plan in self._plans.values():
            for n, i in plan.fin_slot.items():
                if n in fin_feed:
                    rmax = max(rmax, float(plan.fin_amax[i]))
            if plan.act_slot:
                omax = max(omax, float(plan.act_amax.max()))
        for op in self._ops:
            if op["type"] != "conv3":
                continue
            wmax = max(wmax, float(op["conv"].weight.detach().abs().max()))
            for t in op["ins"]:
                bn = getattr(t, "bn", None)
                if bn is not None and t.act != ACT_RELU6:
                    amax = max(amax, float((64.0 * bn.weight.detach().abs() + bn.bias.detach().abs()).max()))
        switched = float(self._range_switched)
        if sync_ranks and torch.distributed.is_available() and torch.distributed.is_initialized() and self._pflat is not None:
            v = torch.tensor([wmax, amax, rmax, omax, switched], dtype=torch.float32, device=self._pflat.device)
            torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX)
            wmax, amax, rmax, omax, switched = (float(x) for x in v.tolist())
        # Round 5: activations can no longer leave the range -- every split convolution scales its fp16 operand from a device-side
        # bound of its sources (h_act_scale in conv_bx3.hip: |gamma| sqrt(n - 1) + |beta| from sc_bn_finalize in training, the
        # recorded maxima for residual sums / in inference), so only the FILTERS (scaled by a fixed 2^8 at pack time) decide `ok`;
        # the activation figures stay in the report (activation_default_scale: the fixed x2 of rounds 1-4 still applies to all).
        #  Inference has no by-construction bound: BatchNorm runs on running statistics and the scales follow the sticky
        # records, which are refreshed at the range_check_every cadence only.  With the cadence OFF (0) nothing would ever notice a
        # clamp, so the observed / residual maxima stay part of `ok` there; `inference_clamped` counts the forwards whose recheck found
        # a tensor beyond the scale that was in force (each of them warned; the last one of a redo chain is NOT repaired).
        act_ok = bool(self.range_check_every) or max(omax, rmax) * 2.0 <= self.FP16_MAX_ACT
        return dict(max_abs_filter=wmax, filter_limit=self.FP16_MAX_WEIGHT, activation_bound=amax, activation_observed=omax,
                    activation_limit=self.FP16_MAX_ACT, residual_absmax=rmax, switched=bool(switched),
                    activation_default_scale=bool(max(omax, rmax) < self.FP16_MAX_ACT / 2),
                    inference_clamped=int(self._inference_clamped), inference_unrepaired=int(self._inference_unrepaired),
                    ok=bool(wmax < self.FP16_MAX_WEIGHT and not switched and act_ok and not self._inference_unrepaired))

    _range_switched = False      # this replica left precision "fp32" because of a range check (reported to the other ranks)
    _inference_clamped = 0       # inference forwards whose range recheck found a clamped operand (each was redone with adapted scales)
    _inference_unrepaired = 0    # ... of which the redo chain gave up (6 re-runs): the returned logits carry clamped operands

    def check_split_range(self, sync_ranks=False):
        """A checkpoint (or a training run) whose filters or activations leave the fp16 range of the default split continues with
        the three-term bf16 split (fp32's exponent range); the warning says how long operands may have been clamped.

        With ``sync_ranks`` the report's all-reduce runs on EVERY rank, also on one that has already switched (or was configured
        with another precision): a rank that skipped it would leave the others waiting in the collective."""
        multi = (sync_ranks and torch.distributed.is_available() and torch.distributed.is_initialized()
                 and torch.distributed.get_world_size() > 1)
        if self.precision != "fp32" and not multi:
            return True          # (the report synchronises the device: only taken where its collective is owed to the other ranks)
        rep = self.split_range_report(sync_ranks)
        if self.precision != "fp32":
            return True
        if not rep["ok"]:
            import warnings
            warnings.warn(f"HyperStarcopUNet: filters outside the range of the two-fp16-term kernels ({rep}); "
                          f"switching to precision='fp32-x3' (three bf16 terms, no range limits)"
                          + ("; another rank of the process group had already switched" if rep["switched"] else ""))
            self.precision = "fp32-x3"
            self._range_switched = True
        return rep["ok"]

    def _record_activation_range(self, plan):
        """inference: one streaming pass per BatchNorm-fed input of a split convolution, raising its sticky record (training steps
        get the same record for free from the BatchNorm-backward reductions)"""
        lib = _lib.load()
        st = stream()
        for name, slot in plan.act_slot.items():
            t = self._tensors[name]
            s = self._src_of(plan, t)
            check(lib.sc_add_srcs_absmax(C.byref(s), None, None, plan.N, t.C, (plan.H >> t.shift) * (plan.W >> t.shift),
                                         plan.act_amax.data_ptr() + 4 * slot, st))

    # -- init conventions of torchvision MobileNetV2 / smp initialize_decoder / initialize_head
    def reset_parameters(self):
        for m in self.encoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
        if self.in_channels != 3:
            self.encoder.features[0][0].reset_parameters()
        for m in self.decoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
        head = self.segmentation_head[0]
        nn.init.xavier_uniform_(head.weight); nn.init.zeros_(head.bias)

    # ------------------------------------------------------------------------------------------
    def _build_ops(self):
        ops, tensors = [], {}
        ir_blocks = self._ir_blocks = {}

        def T(name, Cn, shift, kind, bn=None, act=ACT_NONE):
            t = _T(name, Cn, shift, kind, bn, act)
            tensors[name] = t
            return t

        x = T("x", self.in_channels, 0, "input")
        f = self.encoder.features
        cur = T("f0", 32, 1, "raw", f[0][1], ACT_RELU6)
        ops.append(dict(type="stem", conv=f[0][0], ins=[x], out=cur))
        skips = {}
        for idx in range(1, 18):
            blk = f[idx]
            cin_t = cur
            seq = blk.conv
            j = 0
            h = cur
            if blk.expand:
                e = T(f"f{idx}e", seq[0][0].out_channels, cur.shift, "raw", seq[0][1], ACT_RELU6)
                ops.append(dict(type="pw", conv=seq[0][0], ins=[h], out=e))
                h, j = e, 1
            sh = h.shift + (1 if blk.stride == 2 else 0)
            d = T(f"f{idx}d", seq[j][0].out_channels, sh, "raw", seq[j][1], ACT_RELU6)
            ops.append(dict(type="dw", conv=seq[j][0], stride=blk.stride, ins=[h], out=d))
            p = T(f"f{idx}p", seq[j + 1].out_channels, sh, "raw", seq[j + 2], ACT_NONE)
            ops.append(dict(type="pw", conv=seq[j + 1], ins=[d], out=p))
            if blk.expand:
                ir_blocks[len(ops) - 3] = (len(ops) - 2, len(ops) - 1, blk.stride)      # expand op -> (depthwise op, project op, stride)
            if blk.use_res_connect:
                z = T(f"f{idx}", p.C, sh, "fin")
                ops.append(dict(type="add", ins=[cin_t, p], out=z))
                cur = z
            else:
                cur = p
            if idx in (1, 3, 6, 13):
                skips[idx] = cur
        last = T("f18", 1280, cur.shift, "raw", f[18][1], ACT_RELU6)
        ops.append(dict(type="pw", conv=f[18][0], ins=[cur], out=last))
        cur = last
        skip_list = [skips[13], skips[6], skips[3], skips[1], None]
        for b, blk in enumerate(self.decoder.blocks):
            sh = cur.shift - 1
            ins = [cur] + ([skip_list[b]] if skip_list[b] is not None else [])
            o1 = T(f"d{b}a", blk.conv1[0].out_channels, sh, "raw", blk.conv1[1], ACT_RELU)
            ops.append(dict(type="conv3", conv=blk.conv1[0], ins=ins, up=True, out=o1))
            o2 = T(f"d{b}b", blk.conv2[0].out_channels, sh, "raw", blk.conv2[1], ACT_RELU)
            ops.append(dict(type="conv3", conv=blk.conv2[0], ins=[o1], up=False, out=o2))
            cur = o2
        logits = T("logits", 1, 0, "fin")
        ops.append(dict(type="head", conv=self.segmentation_head[0], ins=[cur], out=logits))
        return ops, tensors

    # ------------------------------------------------------------------------------------------
    # flat parameter / gradient storage (one fused Adam pass, one RCCL all-reduce)
    def _ensure_flat(self):
        params = list(self.parameters())
        dev = params[0].device
        ok = self._pflat is not None and self._pflat.device == dev
        if ok:
            off = 0
            base = self._pflat.data_ptr()
            for p in params:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                off += n
            self._pflat = flat
            self._gflat = torch.zeros(total, dtype=torch.float32, device=dev)
            self._pack_version = None
            self._plans = {}
        return params

    # -- optional per-kernel-family timing (bench.py roofline): events on the launch stream
    profile = None

    profile_detail = False
    _cur_op = ""

    def _pb(self, fam, flop=0.0, nbytes=0.0, flop_exec=None):
        """flop: ALGORITHMIC flops of the launch(es) (the reference's 3x3 convolution); flop_exec: the multiply-adds the kernels
        actually execute when that differs (sub-pixel decoder conv1: 16 instead of 36 taps per low-resolution pixel and up-sampled
        channel), both before the split's products-per-multiply factor"""
        if self.profile is None:
            return None
        if self.profile_detail:
            fam = f"{self._cur_op}|{fam}"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (fam, flop, e0, e1, nbytes, flop if flop_exec is None else flop_exec)

    def _pe(self, tok):
        if tok is not None:
            tok[3].record()
            self.profile.setdefault(tok[0], []).append(tok)

    def collect_profile(self):
        torch.cuda.synchronize()
        out = {}
        for fam, toks in (self.profile or {}).items():
            out[fam] = {"ms": sum(t[2].elapsed_time(t[3]) for t in toks), "flop": sum(t[1] for t in toks), "n": len(toks),
                        "bytes": sum(t[4] for t in toks), "flop_exec": sum(t[5] for t in toks)}
        return out

    _stat_epoch = 0      # bumped whenever parameters / running statistics change through raw pointers (torch's _version does not see those)

    def mark_parameters_changed(self):
        """Call after modifying parameters through raw pointers (fused Adam): packed filters and cached inference constants are stale."""
        self._pack_version = None
        self._stat_epoch += 1

    def flat_parameters(self):
        self._ensure_flat()
        return self._pflat

    def flat_grads(self):
        self._ensure_flat()
        return self._gflat

    def _grad_view(self, p):
        off = (p.data_ptr() - self._pflat.data_ptr()) // 4
        return self._gflat[off:off + p.numel()].view(p.shape)

    # ------------------------------------------------------------------------------------------
    class _Plan:
        pass

    def _get_plan(self, N, H, W, need_grad):
        key = (N, H, W)
        plan = self._plans.get(key)
        dev = self._pflat.device
        if plan is None:
            plan = HyperStarcopUNet._Plan()
            plan.N, plan.H, plan.W = N, H, W
            plan.buf, plan.grad = {}, {}
            f32 = dict(dtype=torch.float32, device=dev)
            lib = _lib.load()
            plan.cst, plan.cstb, plan.stats_v, plan.bsums_v, plan.srows, plan.brows = {}, {}, {}, {}, {}, {}
            kind_of = {"stem": STAT_STEM, "dw": STAT_DW, "conv3": STAT_CONV3, "pw": STAT_CONV1}
            # inverted-residual blocks whose expansion + depthwise pair runs fused in TRAINING at this resolution: expand op -> depthwise op
            plan.irt, plan.irt_of_dw, plan.irt_rows = {}, {}, {}
            if self.fuse_irt:
                for i_e, (i_dw, i_pr, stride) in self._ir_blocks.items():
                    cv, tin = self._ops[i_e]["conv"], self._ops[i_e]["ins"][0]
                    if tin.kind != "input" and _use_irt(N, cv.in_channels, cv.out_channels, H >> tin.shift, W >> tin.shift, stride):
                        plan.irt[i_e], plan.irt_of_dw[i_dw] = i_dw, i_e
                        te, td = self._ops[i_e]["out"], self._ops[i_dw]["out"]
                        plan.irt_rows[te.name] = lib.sc_irt_rows(0, N, H >> tin.shift, W >> tin.shift, stride)
                        plan.irt_rows[td.name] = lib.sc_irt_rows(1, N, H >> tin.shift, W >> tin.shift, stride)
            irt_e = {self._ops[i]["out"].name for i in plan.irt}
            for op in self._ops:
                t = op["out"]
                if t.kind != "input" and t.name not in plan.buf and t.name not in irt_e:      # (a fused block's expanded tensor exists
                    plan.buf[t.name] = torch.empty((N, t.C, H >> t.shift, W >> t.shift), **f32)     #  only in inference: _forward_impl)
                if t.bn is not None:
                    Ho, Wo = H >> t.shift, W >> t.shift
                    # per-work-group partial rows [rows][C][2] (plain stores; summed in fp64 by the finalize kernels)
                    kind = kind_of[op["type"]]
                    if op["type"] == "pw" and _use_pw3(0, N, Ho * Wo, op["conv"].in_channels, op["conv"].out_channels):
                        kind = STAT_PW3
                    elif op["type"] == "pw" and _use_ksplit(N, Ho * Wo, op["conv"].in_channels, op["conv"].out_channels):
                        kind = STAT_CONV1K
                    plan.srows[t.name] = max(lib.sc_stat_rows(kind, N, Ho, Wo), plan.irt_rows.get(t.name, 0))
                    plan.brows[t.name] = lib.sc_stat_rows(STAT_BNBWD, N, Ho, Wo)
                    plan.stats_v[t.name] = torch.empty(plan.srows[t.name] * t.C * 2, **f32)
                    plan.cst[t.name] = torch.zeros((t.C, SC_CST), **f32)
                    plan.cstb[t.name] = torch.zeros((t.C, SC_CST), **f32)
            # inference: stride-1 inverted-residual blocks that run as one launch at this resolution: expand op -> (depthwise op,
            # project op, add op or None); their two 1x1 filters are needed in the PW3 layout (packed for inference forwards only)
            plan.irb = {}
            for i_e, (i_dw, i_pr, stride) in self._ir_blocks.items():
                cv_e, cv_p, tin = self._ops[i_e]["conv"], self._ops[i_pr]["conv"], self._ops[i_e]["ins"][0]
                if tin.kind != "input" and _use_irb(N, cv_e.in_channels, cv_e.out_channels, cv_p.out_channels, H >> tin.shift, W >> tin.shift, stride):
                    i_add = i_pr + 1 if (i_pr + 1 < len(self._ops) and self._ops[i_pr + 1]["type"] == "add"
                                         and self._ops[i_pr + 1]["ins"][1] is self._ops[i_pr]["out"]) else None
                    plan.irb[i_e] = (i_dw, i_pr, i_add)
                    lay = getattr(self, "_irb_layers", None)
                    if lay is None:
                        lay = self._irb_layers = set()
                    if not {i_e, i_pr} <= lay:
                        lay.update((i_e, i_pr))
                        self._pack_version = None
                        self._pack_tables = {}
            # which filter layouts the pointwise layers need at this resolution (sticky over all plans of the network)
            need = getattr(self, "_pw_need", None)
            if need is None:
                need = self._pw_need = {}
            for i, op in enumerate(self._ops):
                if op["type"] == "pw":
                    hw = (H >> op["out"].shift) * (W >> op["out"].shift)
                    for tflip in (0, 1):
                        cv = op["conv"]
                        lay = "pw3" if _use_pw3(tflip, N, hw, cv.in_channels, cv.out_channels) else "mfma"
                        if lay not in need.setdefault((i, tflip), set()):
                            need[(i, tflip)].add(lay)
                            self._pack_version = None
                            self._pack_tables = {}
            fins = [t.name for t in self._tensors.values() if t.kind == "fin" and t.name != "logits"]
            plan.bn_scratch = torch.empty(64 * 2 * max(t.C for t in self._tensors.values()), dtype=torch.float64, device=dev)   # sc_bn_finalize
            plan.fin_slot = {n: i for i, n in enumerate(fins)}
            plan.fin_amax = torch.zeros(len(fins), **f32)        # running max |value| of each residual sum (never lowered)
            raw_feed = self._split_feeders()[0]
            plan.act_slot = {n: i for i, n in enumerate(raw_feed)}
            plan.act_amax = torch.zeros(max(len(raw_feed), 1), **f32)   # sticky max |BN output| of the split kernels' BatchNorm-fed inputs
            plan.n_eval = 0
            plan.has_grad = False
            self._plans[key] = plan
        if need_grad and not plan.has_grad:
            lib = _lib.load()
            f32 = dict(dtype=torch.float32, device=dev)
            ws, up = 0, 0
            n_dw = 0
            irt_e = {self._ops[i]["out"].name for i in plan.irt}
            plan.irt_esums, plan.irt_work = {}, {}
            for i_e in plan.irt:
                cv, tin, te = self._ops[i_e]["conv"], self._ops[i_e]["ins"][0], self._ops[i_e]["out"]
                Hi, Wi = H >> tin.shift, W >> tin.shift
                plan.irt_esums[te.name] = torch.empty(lib.sc_irt_bwd_rows(N, cv.out_channels, Hi, Wi) * cv.out_channels * 2, dtype=torch.float64, device=dev)
                plan.irt_work[te.name] = torch.empty(lib.sc_irt_bwd_workspace_floats(N, cv.in_channels, cv.out_channels, Hi, Wi), **f32)
            for op in self._ops:
                o = op["out"]
                Ho, Wo = H >> o.shift, W >> o.shift
                if op["type"] != "head" and o.kind != "fin" and o.name not in irt_e:
                    plan.grad[o.name] = torch.empty((N, o.C, Ho, Wo), **f32)
                if op["type"] == "add":
                    plan.grad[o.name] = torch.empty((N, o.C, Ho, Wo), **f32)
                conv = op.get("conv")
                if op["type"] in ("pw", "conv3"):
                    ws = max(ws, lib.sc_wgrad_workspace_floats(N, Ho, Wo, conv.out_channels, conv.in_channels,
                                                               conv.kernel_size[0]))
                    if conv.kernel_size[0] == 3:
                        ws = max(ws, lib.sc_wgrad_bx3_workspace_floats(N, Ho, Wo, conv.out_channels, conv.in_channels))
                        if conv.out_channels <= 16 and conv.in_channels in (16, 32):
                            ws = max(ws, lib.sc_wgrad_thin16_workspace_floats(N, Ho, Wo, conv.out_channels, conv.in_channels))
                    if op.get("up"):
                        up = max(up, N * op["ins"][0].C * Ho * Wo)
                elif op["type"] == "stem":
                    ws = max(ws, lib.sc_stem_wgrad_workspace_floats(N, conv.in_channels, H, W))
                elif op["type"] == "head":
                    ws = max(ws, lib.sc_head_wgrad_workspace_floats(N, conv.in_channels, Ho, Wo))
                elif op["type"] == "dw":
                    n_dw += conv.out_channels * 9
            # decoder conv1 weight gradients as box-sum GEMMs (conv_spw.hip): box-sum planes, split source, K-slice partials; the skip
            # channels' dense gradient before it is scattered into its columns
            spw_b, spw_sk = 0, 0
            for op in self._ops:
                if op["type"] == "conv3" and op.get("up"):
                    cv, o_ = op["conv"], op["out"]
                    Hq, Wq = H >> o_.shift, W >> o_.shift
                    cu_ = op["ins"][0].C
                    if _use_spw(N, Hq, Wq, cv.out_channels, cu_):
                        spw_b = max(spw_b, lib.sc_sp_wgrad_workspace_bytes(N, Hq, Wq, cv.out_channels, cu_))
                        if cv.in_channels > cu_:
                            spw_sk = max(spw_sk, cv.out_channels * (cv.in_channels - cu_) * 9)
                            ws = max(ws, lib.sc_wgrad_bx3_workspace_floats(N, Hq, Wq, cv.out_channels, cv.in_channels - cu_),
                                     lib.sc_wgrad_workspace_floats(N, Hq, Wq, cv.out_channels, cv.in_channels - cu_, 3))
            plan.spw_ws = torch.empty(spw_b + 256, dtype=torch.uint8, device=dev) if spw_b else None
            plan.spw_skip = torch.empty(spw_sk, **f32) if spw_sk else None
            plan.ws = torch.empty(ws, **f32)
            plan.ws_floats = ws
            # pointwise weight gradients keep their K-slice partials in buffers of their own until ONE batched reduction at the
            # end of the backward pass (sc_wgrad_reduce_batch) instead of 2-3 dependent few-microsecond launches per layer
            plan.pw_part, plan.pw_table = {}, None
            if self.batch_pw_reduce:
                for i, op in enumerate(self._ops):
                    if op["type"] == "pw" and i not in plan.irt:
                        conv, o = op["conv"], op["out"]
                        Hq, Wq = H >> o.shift, W >> o.shift
                        nfl = (lib.sc_wgrad_pw3_workspace_floats(N, Hq, Wq, conv.out_channels, conv.in_channels)
                               if _use_pw3(2, N, Hq * Wq, conv.in_channels, conv.out_channels)
                               else lib.sc_wgrad_workspace_floats(N, Hq, Wq, conv.out_channels, conv.in_channels, 1))
                        plan.pw_part[i] = torch.empty(nfl, **f32)
            plan.up_tmp = torch.empty(max(up, 1), **f32)
            plan.dw_acc = torch.zeros(n_dw, dtype=torch.float64, device=dev)
            # one float per BatchNorm'd tensor: max |gamma*invstd * g| of its gradient, the range hint of the fp16-split kernels
            plan.gmax = torch.zeros(len(self._ops) + 1, dtype=torch.float32, device=dev)
            for t in self._tensors.values():
                if t.bn is not None:
                    plan.bsums_v[t.name] = torch.empty(plan.brows[t.name] * t.C * 2, dtype=torch.float64, device=dev)
            # inputs of the depthwise convolutions: their BatchNorm-backward sums come out of the fused depthwise backward
            plan.dwrows, plan.dwsums = {}, {}
            for op in self._ops:
                if op["type"] == "dw" and op["ins"][0].bn is not None and op["ins"][0].name not in irt_e:
                    ti = op["ins"][0]
                    plan.dwrows[ti.name] = lib.sc_stat_rows(STAT_DW, N, H >> ti.shift, W >> ti.shift)
                    plan.dwsums[ti.name] = torch.empty(plan.dwrows[ti.name] * ti.C * 2, dtype=torch.float64, device=dev)
            plan.dlogits = torch.empty((N, 1, H, W), **f32)
            plan.has_grad = True
        return plan

    # ------------------------------------------------------------------------------------------
    def _pack_all(self, need_bwd):
        """(Re)pack conv filters into the MFMA kernels' layouts when the parameters changed."""
        lib = _lib.load()
        ver = (tuple(p._version for p in self.parameters()), self._terms, self.split_bf16, self.thin16)    # a precision switch repacks too
        pv = self._pack_version
        need_irb = (not need_bwd) and bool(getattr(self, "_irb_layers", None))      # the fused inference blocks' PW3 layouts
        if pv is not None and pv[0] == ver and (pv[1] or not need_bwd) and (pv[2] or not need_irb):
            return
        st = stream()
        dev = self._pflat.device
        if not hasattr(self, "_wpk"):
            self._wpk = {}
        for i, op in enumerate(self._ops):
            if op["type"] not in ("pw", "conv3"):
                continue
            conv = op["conv"]
            co, ci, ks = conv.out_channels, conv.in_channels, conv.kernel_size[0]
            ent = self._wpk.get(i)
            tf_, tb_ = self._terms
            if ent is None or ent["f"].device != dev or ent["split"] != (self.split_bf16, tf_, tb_):
                cf, cb = _pick_cot(co, ks), _pick_cot(ci, ks)
                if ks == 3 and co <= 16 and ci >= 32 and self.split_bf16:
                    cf = 32       # decoder.blocks.4.conv1 (32 -> 16): the split kernel with half-empty cout blocks still beats
                                  # the fp32-MFMA thin kernel (0.36 vs 0.50 ms); 16 -> 16 layers do not (tools/bench_thin_bx3.py)
                # 3x3 layers with >= 32 output channels run on the 16-bit matrix cores with exactly-split operands
                # (fp32-level accuracy, conv_bx3.hip; `precision` picks the split); thin ones stay on the fp32 MFMA kernels
                xf = self.split_bf16 and ks == 3 and cf >= 32
                xb = self.split_bf16 and ks == 3 and cb >= 32
                nf = lib.sc_packed_weight_floats_bx3(co, ci, cf, 0, tf_) if xf else lib.sc_packed_weight_floats(co, ci, ks, cf, 0)
                nb = lib.sc_packed_weight_floats_bx3(co, ci, cb, 1, tb_) if xb else lib.sc_packed_weight_floats(co, ci, ks, cb, 1)
                ent = dict(cot_f=cf, cot_b=cb, bx3_f=xf, bx3_b=xb, split=(self.split_bf16, tf_, tb_), terms_f=tf_, terms_b=tb_,
                           f=torch.empty(nf, dtype=torch.float32, device=dev),
                           b=torch.empty(nb, dtype=torch.float32, device=dev), tf=None, tb=None, tsd=None)
                # decoder.blocks.4 (<= 16 output channels at full resolution) under the two-fp16-term split: filters in registers
                # (sc_conv3x3_thin16; 0.20-0.28 vs 0.31-0.51 ms per launch).  The backward-data kernel only takes the plain
                # epilogue, so the regular pack is kept beside it.
                if self.split_bf16 and self.thin16 and ks == 3 and len(op["ins"]) == 1:
                    if tf_ == TERMS_F16X2 and co <= 16 and ci in (16, 32):
                        ent["tf"] = torch.empty(lib.sc_packed_weight_floats_thin16(co, ci, 0), dtype=torch.float32, device=dev)
                    if tb_ == TERMS_F16X2 and ci <= 16 and co in (16, 32):
                        ent["tb"] = torch.empty(lib.sc_packed_weight_floats_thin16(co, ci, 1), dtype=torch.float32, device=dev)
                    if tb_ == TERMS_F16X2 and op.get("up") and co == 16 and ci == 32:
                        # decoder.blocks.4.conv1: the half-resolution data gradient in its sub-pixel form (sc_conv3x3_thin16 with down0)
                        ent["tsd"] = torch.empty(lib.sc_packed_weight_floats_thin16(co, ci, 1), dtype=torch.float32, device=dev)
                # decoder conv1 data gradient (up-sampled channels + skip channels): the two outputs as TWO launches with their own
                # cout tiles -- 64-wide for the up-sampled part (a multiple of 64), one 32-wide tile for the <= 32 skip channels --
                # instead of one launch of 32-wide tiles (80 = 64 + 16: 3 x 32; 152 = 128 + 24: 5 x 32): every cout tile stages (loads,
                # BatchNorm-backward prologue, split) the whole gradient patch again.  Measured at 16 x 512^2: decoder.blocks.2.conv1
                # 212 -> 197 us (5 -> 3 passes), decoder.blocks.3.conv1 293 -> 288 us (a 64-wide pass costs two 32-wide ones there: the
                # operand reads and MFMAs, not the staging, are what its work-groups wait for); 288 = 256 + 32, which already runs on
                # 64-wide tiles, loses 5 us to the second launch and stays as it was.  Step +0.6 % (same-box pairs).
                if xb and cb == 32 and op.get("up") and len(op["ins"]) == 2 and self.split_dgrad_launch:
                    cu, cs_ = op["ins"][0].C, op["ins"][1].C
                    if cu % 64 == 0 and cs_ <= 32 and cu + cs_ == ci:
                        ent["bA"] = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, 64, 1, tb_), dtype=torch.float32, device=dev)
                        ent["bB"] = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, 32, 1, tb_), dtype=torch.float32, device=dev)
                        ent["bB_off"] = ent["bB"].numel() // (-(-ci // 32)) * (cu // 32)      # floats before the skip channels' tile
                # decoder conv1 forward as a sub-pixel convolution (the two-fp16-term arithmetic or, "bf16", one bf16 term): phase /
                # parity filters
                if op.get("up") and xf and tf_ in _SP_TERMS and _SP != "0":
                    cu = op["ins"][0].C
                    ent["sp"] = torch.empty(lib.sc_packed_weight_floats_sp(co, cu, ci - cu), dtype=torch.float32, device=dev)
                    ent["sp_cu"] = cu
                # ... and the data gradient of its up-sampled channels (the skip channels' gradient: a 3x3 launch on a 32-wide pack)
                if op.get("up") and xb and tb_ in _SP_TERMS and _SP != "0" and op["ins"][0].C % 32 == 0:
                    cu = op["ins"][0].C
                    ent["spd_vskip"] = bool(ci > cu and lib.sc_spd_vskip_ok(cu, ci - cu))     # decoder.blocks.3: skip gradient in the same launch
                    # ... or as additional channel tiles of that launch (skip tiles), where measured faster than a 3x3 launch of its own
                    ent["spd_stiles"] = bool(ci > cu and not ent["spd_vskip"] and _use_spd_skip_tiles(cu, ci - cu))
                    ent["spd"] = torch.empty(lib.sc_packed_weight_floats_spd(co, cu, ci - cu if ent["spd_stiles"] else 0), dtype=torch.float32, device=dev)
                    ent["sp_cu"] = cu
                    if ci > cu and ent.get("bB") is None and not ent["spd_vskip"] and not ent["spd_stiles"]:
                        ent["bB"] = torch.empty(lib.sc_packed_weight_floats_bx3(co, ci, 32, 1, tb_), dtype=torch.float32, device=dev)
                        ent["bB_off"] = ent["bB"].numel() // (-(-ci // 32)) * (cu // 32)
                self._wpk[i] = ent
                self._pack_tables = {}
            if op["type"] == "pw":       # the split-bf16 layout of sc_conv1x1_pw3, where a plan runs this layer on it
                need = getattr(self, "_pw_need", {})
                for tflip, key in ((0, "pf"), (1, "pb")):
                    if "pw3" in need.get((i, tflip), ()) and (ent.get(key) is None or ent[key].device != dev):
                        ent[key] = torch.empty(lib.sc_packed_weight_floats_pw3(co, ci, tflip), dtype=torch.float32, device=dev)
                        self._pack_tables = {}
                # a fused inference block's filter whose layer does not run on sc_conv1x1_pw3 otherwise: a PW3 pack of its own, filled
                # by inference forwards only (a training step never reads it)
                if (i in getattr(self, "_irb_layers", ()) and "pw3" not in need.get((i, 0), ())
                        and (ent.get("pfi") is None or ent["pfi"].device != dev)):
                    ent["pfi"] = torch.empty(lib.sc_packed_weight_floats_pw3(co, ci, 0), dtype=torch.float32, device=dev)
                    self._pack_tables = {}
        # one launch for all packs: device-side descriptor table, built once per (need_bwd, parameter storage)
        key = (bool(need_bwd), self._pflat.data_ptr(), self._terms, need_irb)
        tab = self._pack_tables.get(key) if hasattr(self, "_pack_tables") else None
        if tab is None:
            import numpy as np
            dt = np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("co_t", "<i4"),
                           ("tflip", "<i4"), ("bx3", "<i4"), ("total", "<u8")])
            # two tables: the layouts the ENCODER's forward reads (pointwise layers, forward direction), and everything else -- the
            # decoder's 3x3 layouts and all backward layouts, most of the work -- which a training step packs on the weight-gradient
            # stream while the encoder runs (see the launches below)
            early, late = [], []

            def add(is_early, w, buf, co, ci, ks, cot, tflip, code):
                total = lib.sc_pack_work_items(co, ci, ks, cot, tflip, int(code))
                (early if is_early else late).append((w.data_ptr(), buf.data_ptr(), co, ci, ks, cot, tflip, int(code), total))

            for i, op in enumerate(self._ops):
                if op["type"] not in ("pw", "conv3"):
                    continue
                conv, ent = op["conv"], self._wpk[i]
                co, ci, ks = conv.out_channels, conv.in_channels, conv.kernel_size[0]
                is_pw = op["type"] == "pw"
                for tflip, buf, cot, bx in ((0, ent["f"], ent["cot_f"], ent["bx3_f"]), (1, ent["b"], ent["cot_b"], ent["bx3_b"])):
                    if tflip and not need_bwd:
                        continue
                    if is_pw and "mfma" not in getattr(self, "_pw_need", {}).get((i, tflip), ("mfma",)):
                        continue        # every plan runs this layer on sc_conv1x1_pw3: the fp32-MFMA layout is not needed
                    add(is_pw and not tflip, conv.weight, buf, co, ci, ks, cot, tflip, (ent["terms_b"] if tflip else ent["terms_f"]) if bx else 0)
                for cot, buf in ((64, ent.get("bA")), (32, ent.get("bB"))):
                    if buf is not None and need_bwd:
                        add(False, conv.weight, buf, co, ci, ks, cot, 1, ent["terms_b"])
                if ent.get("spd") is not None and need_bwd:
                    tfl = (3 if ent["spd_stiles"] else (2 if ent["spd_vskip"] else 1)) | (4 if ent["terms_b"] == 1 else 0)      # | 4: the one-bf16-term layout
                    add(False, conv.weight, ent["spd"], co, ci, ks, ent["sp_cu"], tfl, PACK_SPD)
                if ent.get("sp") is not None:
                    add(False, conv.weight, ent["sp"], co, ci, ks, ent["sp_cu"], 4 if ent["terms_f"] == 1 else 0, PACK_SP)
                for tflip, buf in ((0, ent["tf"]), (1, ent["tb"]), (1, ent.get("tsd"))):
                    if buf is not None and (not tflip or need_bwd):
                        add(False, conv.weight, buf, co, ci, ks, 16, tflip, PACK_THIN16)
                for tflip, buf in ((0, ent.get("pf")), (1, ent.get("pb"))):
                    if buf is not None and (not tflip or need_bwd):
                        add(is_pw and not tflip, conv.weight, buf, co, ci, 1, 0, tflip, PACK_PW3)
                if need_irb and ent.get("pfi") is not None and "pw3" not in getattr(self, "_pw_need", {}).get((i, 0), ()):
                    add(True, conv.weight, ent["pfi"], co, ci, 1, 0, 0, PACK_PW3)

            def table(rows):
                starts, nblk = [], 0
                for r in rows:
                    starts.append(nblk)
                    nblk += -(-r[-1] // 256)
                if not rows:
                    return None
                descs = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).copy()).to(dev)
                return (descs, torch.tensor(starts, dtype=torch.int32).to(dev), len(rows), nblk)
            tab = (table(early), table(late))
            if not hasattr(self, "_pack_tables"):
                self._pack_tables = {}
            self._pack_tables[key] = tab
        # launches: the encoder's layouts on the current stream; the rest, in a training step, on the weight-gradient stream behind a
        # device-scope wait (the parameters are final on the current stream) -- the forward walk waits for it before its first 3x3
        # convolution (_forward_impl), i.e. ~0.12 ms of packing leaves the critical path and runs beside the encoder.  Otherwise
        # (inference, no overlap stream, graph capture) on the current stream too.
        self._late_pack_stream = None
        if tab[0] is not None:
            check(lib.sc_pack_weights_batch(ptr(tab[0][0]), ptr(tab[0][1]), tab[0][2], tab[0][3], st))
        if tab[1] is not None:
            side = None
            if need_bwd and self.overlap_wgrad and self.light_stream_sync and _PACK_SIDE and not torch.cuda.is_current_stream_capturing():
                main = torch.cuda.current_stream()
                if self._side_stream is None or self._side_stream.device != main.device:
                    self._side_stream = torch.cuda.Stream(device=main.device)
                side = self._side_stream
            if side is not None:
                sh = C.c_void_p(side.cuda_stream)
                check(lib.sc_stream_wait_stream(sh, st))
                check(lib.sc_pack_weights_batch(ptr(tab[1][0]), ptr(tab[1][1]), tab[1][2], tab[1][3], sh))
                self._late_pack_stream = sh
            else:
                check(lib.sc_pack_weights_batch(ptr(tab[1][0]), ptr(tab[1][1]), tab[1][2], tab[1][3], st))
        self._pack_version = (ver, bool(need_bwd), need_irb)

    # ------------------------------------------------------------------------------------------
    def _src_of(self, plan, t, up=0, x_cst=None):
        if t.kind == "input":
            if x_cst is None:
                return make_src(plan.buf["x"], t.C, SRC_RAW)
            return make_src(plan.buf["x"], t.C, SRC_NORM, cst=x_cst)
        if t.kind == "raw":
            return make_src(plan.buf[t.name], t.C, SRC_AFFINE, act=t.act, up=up, cst=plan.cst[t.name])
        return make_src(plan.buf[t.name], t.C, SRC_RAW, up=up)

    def _xbound(self, plan, t):
        """device float >= max |activation| of tensor t as the split convolutions stage it, or None (ReLU6 / input: the default scale
        covers it): BatchNorm-fed tensors -> their slot of plan.act_amax (raised by sc_bn_finalize to the by-construction bound in
        training, by the BatchNorm-backward reductions / the streamed record to the observed maximum), residual sums -> plan.fin_amax
        (recorded by sc_add_srcs_absmax in every forward, before any consumer runs)"""
        if t.kind == "fin" and t.name in plan.fin_slot:
            return C.c_void_p(plan.fin_amax.data_ptr() + 4 * plan.fin_slot[t.name])
        slot = plan.act_slot.get(t.name) if t.kind == "raw" else None
        return C.c_void_p(plan.act_amax.data_ptr() + 4 * slot) if slot is not None else None

    def _dy_src(self, plan, t):
        return make_src(plan.grad[t.name], t.C, SRC_BNBWD, act=t.act, cst=plan.cstb[t.name], aux=plan.buf[t.name])

    def _forward_impl(self, x, x_cst, training, need_grad, _recheck=0):
        """x: (N,C,H,W) fp32 device tensor (raw physical units if x_cst is given, else already normalised)."""
        _lib.require_device(x)
        lib = _lib.load()
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"HyperStarcopUNet: expected (N,{self.in_channels},H,W) input, got {tuple(x.shape)}")
        N, _, H, W = x.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"Wrong input shape height={H}, width={W}. Expected image height and width "
                               "divisible by 32.")
        self._ensure_flat()
        x = x.contiguous().float()
        plan = self._get_plan(N, H, W, need_grad)
        plan.buf["x"] = x
        plan.x_cst = x_cst
        plan.generation = getattr(plan, "generation", 0) + 1     # activations of an earlier forward of this shape are gone
        self._pack_all(need_grad)
        st = stream()
        # inference: the BatchNorm constants depend only on parameters and running statistics -- 62 five-microsecond launches of the
        # dependency chain (a tenth of a batch-16 forward) are skipped while neither has changed since this plan last computed them
        eval_key = None
        if not training:
            eval_key = (tuple(p._version for p in self.parameters()), tuple(b._version for b in self.buffers()), self._stat_epoch)
        # (never while a hipGraph is being captured: the graph must contain the finalize launches, or its replays would keep the
        # constants of capture time whatever happens to the parameters afterwards.  Writers that bypass torch's version counters --
        # a broadcast into the flat buffer, raw-pointer updates -- must call mark_parameters_changed().)
        eval_cst_ok = (eval_key is not None and getattr(plan, "eval_cst_key", None) == eval_key
                       and not torch.cuda.is_current_stream_capturing())
        if training:
            self._stat_epoch += 1            # running statistics are about to be updated through raw pointers
            plan.eval_cst_key = None
        if not training and not eval_cst_ok:
            # inference: every BatchNorm's constants up front (they do not depend on the data), so that a fused block can use the
            # constants of tensors it never materialises
            for t in self._tensors.values():
                if t.bn is not None:
                    bn = t.bn
                    check(lib.sc_bn_finalize(None, 0, 1.0, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                                             float(bn.momentum), float(bn.eps), 0, ptr(plan.cst[t.name]), t.C, None, None, st))
        skip = set()
        for i, op in enumerate(self._ops):
            if i in skip:
                continue
            ty, o = op["type"], op["out"]
            Ho, Wo = H >> o.shift, W >> o.shift
            if i in plan.irt:
                if training:
                    self._irt_forward(plan, i, st)
                    skip.add(plan.irt[i])
                    continue
                # inference: the same fused forward with the running-statistics constants (computed up front above), no statistics
                a_irt = self._irt_args(plan, i)
                td = self._ops[plan.irt[i]]["out"]
                self._cur_op = td.name + ":fwd"
                tok = self._pb("k_irt_* (fused expand+dw)")
                check(lib.sc_irt_fwd(C.byref(a_irt), ptr(plan.buf[td.name]), None, st))
                self._pe(tok)
                skip.add(plan.irt[i])
                continue
            if not training and i in plan.irb:
                # inference: the whole inverted-residual block in one launch (conv_irb.hip); covers the depthwise, projection and add ops
                i_dw, i_pr, i_add = plan.irb[i]
                op_d, op_p = self._ops[i_dw], self._ops[i_pr]
                tin, te, td, tp = op["ins"][0], o, op_d["out"], op_p["out"]
                ent_e, ent_p = self._wpk[i], self._wpk[i_pr]
                pe = ent_e["pf"] if "pw3" in self._pw_need.get((i, 0), ()) else ent_e["pfi"]
                pp_ = ent_p["pf"] if "pw3" in self._pw_need.get((i_pr, 0), ()) else ent_p["pfi"]
                a = sc_irb_args()
                a.x = self._src_of(plan, tin)
                a.wpk_expand, a.cst_expand = pe.data_ptr(), plan.cst[te.name].data_ptr()
                a.w_dw, a.cst_dw = op_d["conv"].weight.data_ptr(), plan.cst[td.name].data_ptr()
                a.wpk_project = pp_.data_ptr()
                a.N, a.Cin, a.hidden, a.Cout = N, op["conv"].in_channels, op["conv"].out_channels, op_p["conv"].out_channels
                a.H, a.W, a.stride = H >> tin.shift, W >> tin.shift, self._ir_blocks[i][2]
                if i_add is not None:
                    tz = self._ops[i_add]["out"]
                    a.residual, a.cst_project = 1, plan.cst[tp.name].data_ptr()
                    a.out = plan.buf[tz.name].data_ptr()
                    a.z_absmax = plan.fin_amax.data_ptr() + 4 * plan.fin_slot[tz.name]
                    skip.add(i_add)
                else:
                    a.residual, a.cst_project, a.z_absmax = 0, None, None
                    a.out = plan.buf[tp.name].data_ptr()
                self._cur_op = tp.name + ":fwd"
                tok = self._pb("k_irb (fused block, eval)")
                check(lib.sc_irb_eval(C.byref(a), st))
                self._pe(tok)
                skip.update((i_dw, i_pr))
                continue
            stats = ptr(plan.stats_v[o.name]) if (training and o.bn is not None) else None
            srows = plan.srows.get(o.name, 0)
            bn_done = False        # the producer finalized its BatchNorm itself
            conv = op.get("conv")
            tok = None
            self._cur_op = o.name + ":fwd"
            if self.profile is not None:
                if ty in ("pw", "conv3"):
                    src_elems = sum(t.C * (H >> t.shift) * (W >> t.shift) for t in op["ins"])
                    fl = 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * conv.kernel_size[0] ** 2
                    fle = None
                    ent_ = self._wpk[i]
                    if (op.get("up") and ent_.get("sp") is not None and ent_["terms_f"] in _SP_TERMS and self.split_bf16
                            and ent_["tf"] is None and _use_sp(N, Ho, Wo, o.C)):
                        cu_ = op["ins"][0].C        # per low-resolution pixel: 16 slots per up-sampled channel, 16 per (skip channel, parity)
                        fle = 2.0 * N * (Ho // 2) * (Wo // 2) * 16 * (-(-conv.out_channels // 32) * 32) * (cu_ + 4 * (conv.in_channels - cu_))
                    tok = self._pb("k_conv3_thin_h (fwd+dgrad)" if self._wpk[i]["tf"] is not None else
                                   self._bx3_family("fwd") if self._wpk[i]["bx3_f"] else f"k_conv_mfma<{conv.kernel_size[0]}> (fwd+dgrad)",
                                   fl, 4.0 * (N * src_elems + N * o.C * Ho * Wo + conv.weight.numel()), fle)
                else:
                    tok = self._pb({"stem": "k_stem_*", "dw": "k_dw_*", "head": "k_head_*", "add": "elementwise/bn"}[ty])
            if ty == "stem":
                s = self._src_of(plan, op["ins"][0], x_cst=x_cst)
                check(lib.sc_stem_conv_fwd(C.byref(s), ptr(conv.weight), ptr(plan.buf[o.name]), N, conv.in_channels,
                                           H, W, stats, st))
            elif ty == "dw":
                tin = op["ins"][0]
                s = self._src_of(plan, tin)
                if (training and o.bn is not None and stats is not None and 0 < srows <= _DW_TAIL_ROWS
                        and not torch.cuda.is_current_stream_capturing()):
                    # few statistics rows per channel: the launch finalizes the BatchNorm itself (last arrival per channel by ticket)
                    # -- no dependent ~5 us sc_bn_finalize launch behind it
                    if not hasattr(plan, "bn_tickets"):
                        plan.bn_tickets = {}
                    if o.name not in plan.bn_tickets:
                        plan.bn_tickets[o.name] = torch.zeros(o.C, dtype=torch.int32, device=self._pflat.device)
                    bn = o.bn
                    bt = sc_bn_tail()
                    bt.gamma, bt.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                    bt.running_mean, bt.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    bt.momentum, bt.eps = float(bn.momentum), float(bn.eps)
                    bt.cst, bt.tickets = plan.cst[o.name].data_ptr(), plan.bn_tickets[o.name].data_ptr()
                    xb_ = self._xbound(plan, o)
                    bt.act_bound = xb_.value if xb_ is not None else None
                    check(lib.sc_dwconv3x3_fwd_bn(C.byref(s), ptr(conv.weight), ptr(plan.buf[o.name]), N, o.C,
                                                  H >> tin.shift, W >> tin.shift, op["stride"], stats, C.byref(bt), st))
                    bn_done = True
                else:
                    check(lib.sc_dwconv3x3_fwd(C.byref(s), ptr(conv.weight), ptr(plan.buf[o.name]), N, o.C,
                                               H >> tin.shift, W >> tin.shift, op["stride"], stats, st))
            elif ty in ("pw", "conv3"):
                if ty == "conv3" and self._late_pack_stream is not None:      # the decoder's filter layouts were packed beside the encoder
                    check(lib.sc_stream_wait_stream(st, self._late_pack_stream))
                    self._late_pack_stream = None
                a = sc_conv_args()
                ins = op["ins"]
                a.nsrc = len(ins)
                a.src[0] = self._src_of(plan, ins[0], up=1 if op.get("up") else 0)
                if len(ins) == 2:
                    a.src[1] = self._src_of(plan, ins[1])
                ent = self._wpk[i]
                a.wpk = ent["f"].data_ptr()
                a.N, a.H, a.W, a.Cout = N, Ho, Wo, o.C
                a.ks, a.co_t = conv.kernel_size[0], ent["cot_f"]
                a.out0 = plan.buf[o.name].data_ptr(); a.out1 = None
                a.csplit, a.accum0, a.accum1 = o.C, 0, 0
                a.add0 = None; a.add1 = None
                a.stats = plan.stats_v[o.name].data_ptr() if stats is not None else None
                a.terms = ent["terms_f"]
                if ty == "conv3":
                    for k_, t_ in enumerate(ins):
                        a.xbound[k_] = self._xbound(plan, t_)
                if ty == "pw" and _use_pw3(0, N, Ho * Wo, conv.in_channels, conv.out_channels):
                    fconv = lib.sc_conv1x1_pw3
                    a.wpk = ent["pf"].data_ptr()
                elif (op.get("up") and ent.get("sp") is not None and ent["terms_f"] in _SP_TERMS and self.split_bf16
                      and _use_sp(N, Ho, Wo, o.C)):
                    fconv = lib.sc_conv3x3_sp
                    a.wpk = ent["sp"].data_ptr()
                    srows = lib.sc_sp_stat_rows(N, Ho, Wo, o.C)       # one partial row per work-group tile (fewer than SC_STAT_CONV3's)
                elif ent["tf"] is not None:
                    fconv = lib.sc_conv3x3_thin16
                    a.wpk = ent["tf"].data_ptr()
                elif ent["bx3_f"]:
                    fconv = lib.sc_conv3x3_bx3
                elif _use_ksplit(N, Ho * Wo, conv.in_channels, conv.out_channels, a.ks):
                    fconv = lib.sc_conv1x1_ksplit
                else:
                    fconv = lib.sc_conv2d_mfma
                check(fconv(C.byref(a), st))
            elif ty == "add":
                sa = self._src_of(plan, op["ins"][0])
                sb = self._src_of(plan, op["ins"][1])
                # the residual sums are the only inputs of a split convolution that no BatchNorm bounds (the skips taken after
                # features.3/6/13): their max |value| is recorded for split_range_report
                check(lib.sc_add_srcs_absmax(C.byref(sa), C.byref(sb), ptr(plan.buf[o.name]), N, o.C, Ho * Wo,
                                             plan.fin_amax.data_ptr() + 4 * plan.fin_slot[o.name], st))
            elif ty == "head":
                s = self._src_of(plan, op["ins"][0])
                check(lib.sc_head_conv_fwd(C.byref(s), ptr(conv.weight), ptr(conv.bias), ptr(plan.buf[o.name]),
                                           N, conv.in_channels, Ho, Wo, st))
            self._pe(tok)
            if o.bn is not None and training and not bn_done and not ("f" in _EXP_NO_BNFIN and plan.generation > 3):
                bn = o.bn
                check(lib.sc_bn_finalize(stats, srows, float(N * Ho * Wo), ptr(bn.weight), ptr(bn.bias),
                                         ptr(bn.running_mean), ptr(bn.running_var), float(bn.momentum), float(bn.eps),
                                         1 if training else 0, ptr(plan.cst[o.name]), o.C, ptr(plan.bn_scratch) if _BN_PRE else None,
                                         self._xbound(plan, o), st))
        if self._late_pack_stream is not None:      # (no 3x3 layer ran: still order the main stream after the pack)
            check(lib.sc_stream_wait_stream(st, self._late_pack_stream))
            self._late_pack_stream = None
        if training:      # one multi-tensor launch for the 62 step counters
            torch._foreach_add_(self._nbt_list(), 1)
        else:
            plan.eval_cst_key = eval_key
        plan.training = training
        if (not training and self.precision == "fp32" and self.range_check_every and plan.act_slot
                and not torch.cuda.is_current_stream_capturing()):
            # inference: BatchNorm uses running statistics, so no bound of its output exists before the data has flowed.  The split
            # convolutions scale their operands from the sticky records of the observed maxima (plan.act_amax); at the check cadence
            # (first forward of a shape included) the handful of tensors is streamed once more and, if a value was beyond what the
            # scale in force during THIS forward could carry (it was clamped), the forward is redone -- the records are updated by
            # then, so the scales adapt; the precision mode does not change.  (Training needs none of this: sc_bn_finalize leaves
            # the by-construction bound before any consumer runs.)
            plan.n_eval += 1
            if _recheck or plan.n_eval % self.range_check_every == 1 or self.range_check_every == 1:
                before = plan.act_amax.clone()
                self._record_activation_range(plan)
                used = torch.where(before * 2 > self.FP16_MAX_ACT,
                                   torch.exp2(torch.floor(torch.log2(self.FP16_MAX_ACT / before.clamp_min(1e-30)))), torch.full_like(before, 2.0))
                if bool((plan.act_amax * used > 2 * self.FP16_MAX_ACT).any()):
                    import warnings
                    worst = float((plan.act_amax * used).max()) / 2.0
                    if _recheck < 6:
                        if _recheck == 0:
                            self._inference_clamped += 1
                            warnings.warn(f"HyperStarcopUNet (inference, precision='fp32'): an activation of {worst:.4g} (in units of the "
                                          f"operand scale in force) exceeded the fp16 range of the split convolutions "
                                          f"({self.FP16_MAX_ACT:g}) and was clamped; the forward is redone with the adapted scale.  "
                                          f"Forwards between two checks (range_check_every={self.range_check_every}) are not re-examined: "
                                          f"set range_check_every=1 for data of unknown range")
                        return self._forward_impl(x, x_cst, training, need_grad, _recheck=_recheck + 1)
                    self._inference_unrepaired += 1
                    warnings.warn(f"HyperStarcopUNet (inference, precision='fp32'): activations still outside the fp16 range after 6 "
                                  f"re-runs with adapted scales ({worst:.4g}); the returned logits carry operands clamped to "
                                  f"+-65504/scale.  split_range_report()['ok'] is now False; use precision='fp32-x3' for this input")
        return plan

    def _irt_args(self, plan, i_e):
        i_dw = plan.irt[i_e]
        op, cv_d = self._ops[i_e], self._ops[i_dw]["conv"]
        cv, tin, te = op["conv"], op["ins"][0], op["out"]
        a = sc_irt_args()
        a.x = self._src_of(plan, tin)
        a.w_expand, a.w_dw, a.cst_expand = cv.weight.data_ptr(), cv_d.weight.data_ptr(), plan.cst[te.name].data_ptr()
        a.N, a.Cin, a.hidden = plan.N, cv.in_channels, cv.out_channels
        a.H, a.W, a.stride = plan.H >> tin.shift, plan.W >> tin.shift, self._ops[i_dw]["stride"]
        return a

    def _irt_forward(self, plan, i_e, st):
        """training forward of a fused block: statistics of the (never stored) expanded tensor -> its BatchNorm constants -> raw
        depthwise output + statistics -> the depthwise BatchNorm's constants (conv_irt.hip)"""
        lib = _lib.load()
        i_dw = plan.irt[i_e]
        te, td = self._ops[i_e]["out"], self._ops[i_dw]["out"]
        a = self._irt_args(plan, i_e)
        cnt_e = float(plan.N * a.H * a.W)
        Hd_, Wd_ = plan.H >> td.shift, plan.W >> td.shift
        self._cur_op = te.name + ":fwd"
        tok = self._pb("k_irt_* (fused expand+dw)")
        check(lib.sc_irt_expand_stats(C.byref(a), ptr(plan.stats_v[te.name]), st))
        self._pe(tok)
        bn = te.bn
        check(lib.sc_bn_finalize(ptr(plan.stats_v[te.name]), plan.irt_rows[te.name], cnt_e, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                 ptr(bn.running_var), float(bn.momentum), float(bn.eps), 1, ptr(plan.cst[te.name]), te.C,
                                 ptr(plan.bn_scratch) if _BN_PRE else None, None, st))
        self._cur_op = td.name + ":fwd"
        tok = self._pb("k_irt_* (fused expand+dw)")
        check(lib.sc_irt_fwd(C.byref(a), ptr(plan.buf[td.name]), ptr(plan.stats_v[td.name]), st))
        self._pe(tok)
        bn = td.bn
        check(lib.sc_bn_finalize(ptr(plan.stats_v[td.name]), plan.irt_rows[td.name], float(plan.N * Hd_ * Wd_), ptr(bn.weight), ptr(bn.bias),
                                 ptr(bn.running_mean), ptr(bn.running_var), float(bn.momentum), float(bn.eps), 1, ptr(plan.cst[td.name]), td.C,
                                 ptr(plan.bn_scratch) if _BN_PRE else None, None, st))

    def _nbt_list(self):
        nbt = getattr(self, "_nbt", None)
        if nbt is None or nbt[0].device != self._pflat.device:
            nbt = [t.bn.num_batches_tracked for t in self._tensors.values() if t.bn is not None]
            self._nbt = nbt
        return nbt

    # weight gradients are off the critical path (nothing in the backward pass consumes them): they run on a second
    # HIP stream, forked after each layer's BatchNorm-backward constants and joined before the optimiser, so they fill
    # the CUs the small dgrad / reduce kernels of the dependency chain leave idle.
    overlap_wgrad = True
    batch_pw_reduce = True      # one reduction launch for all pointwise weight gradients (False: per-layer reductions; test switch)
    fuse_dw_bwd = True          # depthwise dgrad + wgrad + the input's BatchNorm-backward sums in one kernel (False: the three separate kernels)
    fuse_irt = _IRT != "0"      # training: expansion + depthwise of the stride-2 blocks the rule picks without the 6x tensor (conv_irt.hip)
    fuse_head_bn = True         # BatchNorm-backward sums of the decoder's last tensor in the head backward
    # ... and of the <= 32 x 32 depthwise outputs in the projections' data-gradient epilogues (sc_conv1x1_pw3 + sc_bnr_args; VERDICT r5 #1b).
    # Built and parity-tested in round 6, measured LEVEL on the step (1481.8 / 1482.0 vs 1481.5 / 1481.1 tiles/s, tools/ab_switch.sh pw_bnr:
    # the pointwise data gradients get 0.06 ms slower, the BatchNorm launches 0.04 ms faster) -- off
    pw_bnr = False
    batch_dw_cast = True        # the depthwise filter gradients' fp64 -> fp32 rounding in one launch per walk (17 forks fewer)
    light_stream_sync = True    # fork points of the weight-gradient stream: events without the system-scope fence (sc_stream_wait_stream)
    split_dgrad_launch = True   # decoder conv1 data gradient: up-sampled and skip channels as two launches with their own cout tiles
    thin16 = True               # decoder.blocks.4 on sc_conv3x3_thin16 (0: the fp32-MFMA thin kernels)
    split_bf16 = True        # 3x3 convs with >= 32 output channels on the 16-bit matrix cores (False: everything on the fp32 MFMA)
    bn_small_max = 16384     # BatchNorm backward in one launch (block per channel) when a channel has at most this many elements
    # Arithmetic of the 3x3 convolutions with >= 32 channels (conv_bx3.hip): every fp32 operand is split exactly into a few
    # 16-bit terms while it is staged and the leading products are accumulated in fp32 on the matrix cores.
    #   "fp32"       two fp16 terms per operand (22 significand bits, exact power-of-two range scaling), three products:
    #                error below the fp32 accumulation error of the reduction -- the parity / bench default
    #   "fp32-x3"    three bf16 terms, six products (error = one fp32 rounding, fp32's full exponent range; 11 % slower)
    #   "fp32-bwd2"  three bf16 terms forward, two in dgrad/wgrad (operand error 2^-18; logits identical to "fp32-x3")
    #   "fp32-2"     two bf16 terms everywhere (logits ~2e-5 from the oracle, inside the 1e-4 contract with less margin)
    #   "bf16"       one bf16 term: plain bf16 matrix math, fp32 accumulation and fp32 tensors in HBM (BASELINE configs[3])
    precision = "fp32"
    _TERMS = {"fp32": (TERMS_F16X2, TERMS_F16X2), "fp32-x3": (3, 3), "fp32-bwd2": (3, 2), "fp32-2": (2, 2), "bf16": (1, 1)}

    def _bx3_family(self, which):
        """profiling family of the split-bf16 3x3 kernel: forward and dgrad launches are one family unless they run with a
        different number of terms (then their MFMA ceilings differ)"""
        tf_, tb_ = self._terms
        return "k_conv3_bx3 (fwd+dgrad)" if tf_ == tb_ else f"k_conv3_bx3 ({which})"

    @property
    def _terms(self):
        """(forward, backward) bf16 terms per operand"""
        if self.precision not in self._TERMS:
            raise ValueError(f"precision must be one of {sorted(self._TERMS)} (got {self.precision!r})")
        return self._TERMS[self.precision]
    _side_stream = None
    _late_pack_stream = None

    def _backward_impl(self, plan, dlogits, on_tail_ready=None, only_ops=None, exchange_follows=None):
        """Fills the flat gradient buffer from dL/dlogits.  Needs the plan of a training-mode forward.

        ``only_ops`` (a list of indices into ``self._ops``; parity tests): run the backward of just these ops, each from whatever
        ``plan.grad[out]`` / ``plan.buf`` / ``plan.cst`` hold -- the teacher-forced per-layer gates feed every layer the oracle's
        activations and upstream gradient through exactly the launches a training step makes for it.

        ``exchange_follows``: a gradient exchange (any collective that may leave the device) reads the flat gradient buffer right
        after this call -- the final join of the weight-gradient stream then uses a default (system-fence) event; default: whether
        ``on_tail_ready`` is given.  Callers with a non-bucketed exchange (a grad_sync object without begin()) pass True.

        ``on_tail_ready(lo, hi)`` is called once, when the walk leaves the decoder: every gradient of the decoder and head
        parameters (flat range [lo, hi), two thirds of the buffer) has been queued, so a data-parallel caller can start
        reducing that bucket while the encoder's backward still runs.  It is called with the weight-gradient stream
        current (ordered after the main stream), i.e. a collective issued inside it waits for both."""
        lib = _lib.load()
        st = stream()
        main = torch.cuda.current_stream()
        side = None
        if self.overlap_wgrad:
            if self._side_stream is None or self._side_stream.device != main.device:
                self._side_stream = torch.cuda.Stream(device=main.device)
            side = self._side_stream

        # experiment (STARCOP_EXP_SIDE2=1): the pointwise weight gradients (own partial buffers) on a SECOND weight-gradient stream
        side2 = None
        if side is not None and _EXP_SIDE2:
            if getattr(self, "_side_stream2", None) is None or self._side_stream2.device != main.device:
                self._side_stream2 = torch.cuda.Stream(device=main.device)
            side2 = self._side_stream2
        hnd = {id(s_): C.c_void_p(s_.cuda_stream) for s_ in (main, side, side2) if s_ is not None}      # raw handles, looked up once per walk

        def wait_stream(waiter, signaller):
            # same-device ordering without the system-scope fence of a default event (sc_stream_wait_stream)
            if self.light_stream_sync:
                check(lib.sc_stream_wait_stream(hnd[id(waiter)], hnd[id(signaller)]))
            else:
                waiter.wait_stream(signaller)

        def wgrad_launch(fn, second=False):
            """run fn(stream_handle) on the weight-gradient stream, ordered after everything queued on the main stream.  (One fork per
            launch: serving two to eight launches with one fork -- fewer markers in the main queue -- measured 0.5-2.5 % SLOWER, the
            weight gradients then start too late to fill the gaps of the data-gradient chain.)"""
            if _EXP_NO_WGRAD:
                return           # elimination experiment (results wrong): what the step costs without any weight-gradient launch
            if side is None:
                fn(st)
            else:
                sd = side2 if (second and side2 is not None) else side
                wait_stream(sd, main)
                fn(hnd[id(sd)])       # (every fn launches through the C ABI with the handle it is given: no stream context switch needed)

        N, H, W = plan.N, plan.H, plan.W
        if not getattr(plan, "training", False):
            raise RuntimeError("HyperStarcopUNet.backward: gradients need a train-mode forward (BatchNorm batch "
                               "statistics); call .train() first")
        dlogits = dlogits.contiguous()
        plan.dw_acc.zero_()
        half_bwd = self._terms[1] == TERMS_F16X2
        gmax_slot = {}
        if half_bwd:
            plan.gmax.zero_()
        written = set()
        res_of = {}       # tensor name -> name of the residual sum z (z = t + ...)
        for op in self._ops:
            if op["type"] == "add":
                res_of[op["ins"][0].name] = op["out"].name
        gv = self._grad_view
        dw_off = 0
        dw_offs = {}
        for i, op in enumerate(self._ops):
            if op["type"] == "dw":
                dw_offs[i] = dw_off
                dw_off += op["conv"].out_channels * 9

        reduced = set()      # tensors whose BatchNorm-backward sums were produced by the launch that wrote their gradient
        reduced32 = {}       # ... by a convolution data-gradient launch (sc_bnr_args): name -> (float rows, number of rows)
        n_cons = {}          # consumers per tensor: a gradient is complete after ONE launch only for single-consumer tensors
        for op in self._ops:
            for t in op["ins"]:
                n_cons[t.name] = n_cons.get(t.name, 0) + 1
        prod_idx = {op["out"].name: k for k, op in enumerate(self._ops)}      # producer op of every tensor (its range-hint slot)
        pw_pending = []      # pointwise weight gradients whose K-slice partials await the batched reduction

        def bn_backward(t, slot=None):
            Ho, Wo = H >> t.shift, W >> t.shift
            amax = None
            if slot is not None:
                amax = plan.gmax.data_ptr() + 4 * slot
                gmax_slot[t.name] = amax
            if t.name in reduced32:        # the data-gradient launch that wrote this gradient left float rows (sc_bnr_args)
                rows_t, nrows = reduced32[t.name]
                check(lib.sc_bn_bwd_finalize_rows32(ptr(rows_t), nrows, float(N * Ho * Wo), ptr(plan.cst[t.name]),
                                                    ptr(gv(t.bn.weight)), ptr(gv(t.bn.bias)), ptr(plan.cstb[t.name]), t.C,
                                                    ptr(plan.bn_scratch) if _BN_PRE else None, st))
                return
            if t.name in reduced:          # the launch that wrote this gradient left the sums (and raised the range-hint slot)
                if "b" in _EXP_NO_BNFIN and plan.generation > 3:
                    return
                check(lib.sc_bn_bwd_finalize(ptr(plan.dwsums[t.name]), plan.dwrows[t.name], float(N * Ho * Wo), ptr(plan.cst[t.name]),
                                             ptr(gv(t.bn.weight)), ptr(gv(t.bn.bias)), ptr(plan.cstb[t.name]), t.C, st))
                return
            aslot = plan.act_slot.get(t.name)           # BatchNorm-fed input of a split convolution: sticky max |BN(y)| record
            aact = plan.act_amax.data_ptr() + 4 * aslot if aslot is not None else None
            if N * Ho * Wo <= self.bn_small_max and t.C >= 64:        # low-resolution layers: one launch, one block per channel
                check(lib.sc_bn_bwd_small(ptr(plan.grad[t.name]), ptr(plan.buf[t.name]), ptr(plan.cst[t.name]), t.act, N, t.C,
                                          Ho * Wo, ptr(gv(t.bn.weight)), ptr(gv(t.bn.bias)), ptr(plan.cstb[t.name]), amax, aact, st))
                return
            check(lib.sc_bn_bwd_reduce(ptr(plan.grad[t.name]), ptr(plan.buf[t.name]), ptr(plan.cst[t.name]), t.act,
                                       ptr(plan.bsums_v[t.name]), N, t.C, Ho * Wo, amax, aact, st))
            if "b" in _EXP_NO_BNFIN and plan.generation > 3:
                return
            check(lib.sc_bn_bwd_finalize(ptr(plan.bsums_v[t.name]), plan.brows[t.name], float(N * Ho * Wo), ptr(plan.cst[t.name]),
                                         ptr(gv(t.bn.weight)), ptr(gv(t.bn.bias)), ptr(plan.cstb[t.name]), t.C, st))

        def bnr_for(t, nrows, enabled=_BNR):
            """sc_bnr_args for the data-gradient launch that is about to write the COMPLETE gradient of tensor t (a single-consumer,
            BatchNorm'd tensor that nothing wrote or will add to): the launch leaves t's BatchNorm-backward sums (nrows partial rows) and
            range hint, instead of sc_bn_bwd_reduce streaming (gradient, y) again -- 0.41 ms of such passes per batch-16 step in the decoder"""
            if (not enabled or t.bn is None or t.kind != "raw" or n_cons.get(t.name, 0) != 1 or t.name in written
                    or res_of.get(t.name) is not None):
                return None
            if not hasattr(plan, "bnr_rows"):
                plan.bnr_rows = {}
            key = (t.name, nrows)
            if key not in plan.bnr_rows:
                plan.bnr_rows[key] = torch.empty(nrows * t.C * 2, dtype=torch.float32, device=self._pflat.device)
            b = sc_bnr_args()
            b.y, b.cst, b.act = plan.buf[t.name].data_ptr(), plan.cst[t.name].data_ptr(), t.act
            b.rows = plan.bnr_rows[key].data_ptr()
            b.absmax = (plan.gmax.data_ptr() + 4 * prod_idx[t.name]) if (half_bwd and self._ops[prod_idx[t.name]]["type"] == "conv3") else None
            reduced32[t.name] = (plan.bnr_rows[key], nrows)
            return b

        def reduce_pointwise_batch():
            """ONE launch (weight-gradient stream) sums the K-slice partials of every pointwise weight gradient queued so far"""
            if not pw_pending:
                return
            raw = b"".join(bytes(p_) for p_ in pw_pending)
            if plan.pw_table is None:
                plan.pw_table = {}
            if raw not in plan.pw_table:
                # descriptors are static for a plan, so the device table is built once -- and another one when a descriptor changes
                # (parameters re-flattened / moved, a partial backward of the parity tests); earlier tables stay alive because a
                # launch on the weight-gradient stream may still be reading them
                import numpy as np
                starts, nblk = [], 0
                for p_ in pw_pending:
                    starts.append(nblk)
                    nblk += -(-int(p_.total) // 256)
                plan.pw_table[raw] = (torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(self._pflat.device),
                                      torch.tensor(starts, dtype=torch.int32).to(self._pflat.device), len(pw_pending), nblk)
            tab = plan.pw_table[raw]
            tok = self._pb("k_wgrad_mfma<1> (+reduce)")
            wgrad_launch(lambda sx: check(lib.sc_wgrad_reduce_batch(ptr(tab[0]), ptr(tab[1]), tab[2], tab[3], sx)), second=True)
            self._pe(tok)
            pw_pending.clear()

        dw_casts = []        # (fp64 accumulator, flat fp32 gradient view, n) of the fused depthwise backward launches walked so far
        last_dw = min((k for k, op_ in enumerate(self._ops) if op_["type"] == "dw"), default=-1)

        def cast_dw_gradients():
            """ONE launch (weight-gradient stream) rounds every depthwise filter gradient accumulated so far to fp32 -- per layer it was
            a 5 us kernel behind its own fork of the weight-gradient stream, i.e. 17 more markers in the main queue per step"""
            if not dw_casts:
                return
            key = tuple(dw_casts)
            if getattr(plan, "dw_cast_table", None) is None:
                plan.dw_cast_table = {}
            if key not in plan.dw_cast_table:
                import numpy as np
                arr = np.array([(a, b, n) for a, b, n in dw_casts], dtype=np.uint64)
                plan.dw_cast_table[key] = (torch.from_numpy(arr.view(np.uint8).copy()).to(self._pflat.device), len(dw_casts))
            tab = plan.dw_cast_table[key]
            wgrad_launch(lambda sx, tab=tab: check(lib.sc_cast_f64_f32_batch(ptr(tab[0]), tab[1], sx)))
            dw_casts.clear()

        last_pw = min((k for k, op_ in enumerate(self._ops) if op_["type"] == "pw" and k in plan.pw_part), default=-1)
        tail_lo = sum(p.numel() for p in self.encoder.parameters())
        tail_pending = on_tail_ready is not None
        irt_done = set()
        for i in (range(len(self._ops) - 1, -1, -1) if only_ops is None else only_ops):
            if i in irt_done:
                continue
            op = self._ops[i]
            ty, o = op["type"], op["out"]
            Ho, Wo = H >> o.shift, W >> o.shift
            conv = op.get("conv")
            if tail_pending and (conv is None or (conv.weight.data_ptr() - self._pflat.data_ptr()) // 4 < tail_lo):
                tail_pending = False            # first encoder op: the decoder/head bucket is complete
                if side is not None:
                    side.wait_stream(main)      # (a default event: the collective that starts here may leave the device)
                    with torch.cuda.stream(side):
                        on_tail_ready(tail_lo, self._gflat.numel())
                else:
                    on_tail_ready(tail_lo, self._gflat.numel())
            self._cur_op = o.name + ":bwd"
            if ty == "head":
                tin = op["ins"][0]
                s = self._src_of(plan, tin)
                tok = self._pb("k_head_*")
                if conv.in_channels == 16 and _HEAD_FUSED_BWD:
                    # one sweep over (dlogits, x) for gin, dW and dbias (sc_head_conv_bwd) -- and, the head being the only consumer
                    # of the decoder's last tensor, that tensor's BatchNorm-backward sums and range hint (its raw values stream
                    # through the kernel anyway): saves sc_bn_bwd_reduce's pass over 2 x 268 MB at 16 x 512^2
                    bns = bna = None
                    if self.fuse_head_bn and tin.bn is not None and tin.kind == "raw" and n_cons.get(tin.name, 0) == 1:
                        if tin.name not in plan.dwsums:
                            plan.dwrows[tin.name] = lib.sc_head_bwd_bn_rows(N, Ho, Wo)
                            plan.dwsums[tin.name] = torch.empty(plan.dwrows[tin.name] * tin.C * 2, dtype=torch.float64, device=self._pflat.device)
                        bns = ptr(plan.dwsums[tin.name])
                        if half_bwd:
                            bna = plan.gmax.data_ptr() + 4 * prod_idx[tin.name]
                        reduced.add(tin.name)
                    check(lib.sc_head_conv_bwd(ptr(dlogits), C.byref(s), ptr(conv.weight), ptr(plan.grad[tin.name]), ptr(plan.ws),
                                               plan.ws_floats, ptr(gv(conv.weight)), ptr(gv(conv.bias)), N, conv.in_channels, Ho, Wo,
                                               bns, bna, st))
                else:
                    wgrad_launch(lambda sx, s=s, conv=conv, Ho=Ho, Wo=Wo: check(lib.sc_head_conv_wgrad(
                        ptr(dlogits), C.byref(s), ptr(plan.ws), plan.ws_floats, ptr(gv(conv.weight)), ptr(gv(conv.bias)), N,
                        conv.in_channels, Ho, Wo, sx)))
                    check(lib.sc_head_conv_dgrad(ptr(dlogits), ptr(conv.weight), ptr(plan.grad[tin.name]), N,
                                                 conv.in_channels, Ho, Wo, st))
                self._pe(tok)
                written.add(tin.name)
                continue
            if ty == "add":
                # z = a + BN(p): dL/d(BN(p)) = dL/dz (alias); a receives dL/dz through its expand-conv dgrad epilogue
                p_t = op["ins"][1]
                plan.grad[p_t.name] = plan.grad[o.name]
                written.add(p_t.name)
                continue
            tok = self._pb("elementwise/bn")
            bn_backward(o, i if (half_bwd and ty == "conv3") else None)
            self._pe(tok)
            dy = self._dy_src(plan, o)
            if ty == "stem":
                s = self._src_of(plan, op["ins"][0], x_cst=plan.x_cst)
                tok = self._pb("k_stem_*")
                wgrad_launch(lambda sx, dy=dy, s=s, conv=conv: check(lib.sc_stem_conv_wgrad(
                    C.byref(dy), C.byref(s), ptr(plan.ws), plan.ws_floats, ptr(gv(conv.weight)), N, conv.in_channels, H, W, sx)))
                self._pe(tok)
                continue
            if ty == "dw" and i in plan.irt_of_dw:
                # fused block: ONE sweep over (dy_d, x) for the depthwise filter gradient, the BatchNorm-backward sums of the
                # expanded tensor and the part of dx that does not depend on them; the constants; the Cin -> Cin fix-up of dx; the
                # expansion filter's gradient from the partial rows on the weight-gradient stream.  Covers the expand op too.
                i_e = plan.irt_of_dw[i]
                irt_done.add(i_e)
                op_e = self._ops[i_e]
                te, tin, cv_e = op_e["out"], op_e["ins"][0], op_e["conv"]
                a_irt = self._irt_args(plan, i_e)
                acc = plan.dw_acc[dw_offs[i]:dw_offs[i] + o.C * 9]
                work, esums = plan.irt_work[te.name], plan.irt_esums[te.name]
                tok = self._pb("k_irt_* (fused expand+dw)")
                wgrad_launch(lambda sx, a_irt=a_irt, work=work: check(lib.sc_irt_xmoments(C.byref(a_irt), ptr(work), sx)))
                check(lib.sc_irt_bwd(C.byref(a_irt), C.byref(dy), ptr(esums), ptr(acc), ptr(work), st))
                if self.batch_dw_cast:
                    dw_casts.append((acc.data_ptr(), gv(conv.weight).data_ptr(), o.C * 9))
                else:
                    wgrad_launch(lambda sx, acc=acc, conv=conv, o=o: check(lib.sc_cast_f64_f32(ptr(acc), ptr(gv(conv.weight)), o.C * 9, sx)))
                check(lib.sc_bn_bwd_finalize(ptr(esums), lib.sc_irt_bwd_rows(N, te.C, a_irt.H, a_irt.W), float(N * a_irt.H * a_irt.W),
                                             ptr(plan.cst[te.name]), ptr(gv(te.bn.weight)), ptr(gv(te.bn.bias)), ptr(plan.cstb[te.name]), te.C, st))
                z = res_of.get(tin.name)
                check(lib.sc_irt_bwd_fix(C.byref(a_irt), ptr(plan.cstb[te.name]), ptr(work), ptr(plan.grad[tin.name]),
                                         ptr(plan.grad[z]) if z is not None else None, 1 if tin.name in written else 0, st))
                wgrad_launch(lambda sx, a_irt=a_irt, work=work, te=te, cv_e=cv_e: check(lib.sc_irt_wgrad_finalize(
                    C.byref(a_irt), ptr(plan.cstb[te.name]), ptr(work), ptr(gv(cv_e.weight)), sx)))
                self._pe(tok)
                written.add(tin.name)
                if i == last_dw:
                    cast_dw_gradients()
                continue
            if ty == "dw":
                tin = op["ins"][0]
                Hi, Wi = H >> tin.shift, W >> tin.shift
                s = self._src_of(plan, tin)
                acc = plan.dw_acc[dw_offs[i]:dw_offs[i] + o.C * 9]
                tok = self._pb("k_dw_*")
                if (self.fuse_dw_bwd and tin.name not in written and tin.bn is not None and tin.name in plan.dwsums
                        and n_cons.get(tin.name, 0) == 1):      # the fused sums are complete only for a single-consumer input
                    # one pass: dx, dW and the BatchNorm-backward sums of the (6x expanded) input tensor
                    check(lib.sc_dwconv3x3_bwd_fused(C.byref(dy), C.byref(s), ptr(conv.weight), ptr(plan.grad[tin.name]), ptr(acc),
                                                     ptr(plan.dwsums[tin.name]), N, o.C, Hi, Wi, op["stride"], st))
                    if self.batch_dw_cast:
                        dw_casts.append((acc.data_ptr(), gv(conv.weight).data_ptr(), o.C * 9))
                    else:
                        wgrad_launch(lambda sx, acc=acc, conv=conv, o=o: check(lib.sc_cast_f64_f32(ptr(acc), ptr(gv(conv.weight)), o.C * 9, sx)))
                    self._pe(tok)
                    written.add(tin.name)
                    reduced.add(tin.name)
                    if i == last_dw:
                        cast_dw_gradients()
                    continue

                def dw_wgrad(sx, dy=dy, s=s, acc=acc, conv=conv, o=o, Hi=Hi, Wi=Wi, stride=op["stride"]):
                    check(lib.sc_dwconv3x3_wgrad(C.byref(dy), C.byref(s), ptr(acc), N, o.C, Hi, Wi, stride, sx))
                    check(lib.sc_cast_f64_f32(ptr(acc), ptr(gv(conv.weight)), o.C * 9, sx))
                wgrad_launch(dw_wgrad)
                check(lib.sc_dwconv3x3_dgrad(C.byref(dy), ptr(conv.weight), ptr(plan.grad[tin.name]),
                                             1 if tin.name in written else 0, N, o.C, Hi, Wi, op["stride"], st))
                self._pe(tok)
                written.add(tin.name)
                if i == last_dw:
                    cast_dw_gradients()
                continue
            # pw / conv3 : weight gradient
            ins = op["ins"]
            ks = conv.kernel_size[0]
            wa = sc_wgrad_args()
            wa.dy = dy
            wa.nsrc = len(ins)
            wa.src[0] = self._src_of(plan, ins[0], up=1 if op.get("up") else 0)
            if len(ins) == 2:
                wa.src[1] = self._src_of(plan, ins[1])
            wa.N, wa.H, wa.W, wa.Cout, wa.Cin, wa.ks = N, Ho, Wo, o.C, conv.in_channels, ks
            wa.part = plan.ws.data_ptr(); wa.part_floats = plan.ws_floats
            wa.dw = gv(conv.weight).data_ptr()
            wa.terms = self._terms[1]
            wa.absmax = gmax_slot.get(o.name)
            if ty == "conv3":
                for k_, t_ in enumerate(ins):
                    wa.xbound[k_] = self._xbound(plan, t_)
            flop = 2.0 * N * Ho * Wo * conv.out_channels * conv.in_channels * ks * ks
            if (op.get("up") and plan.spw_ws is not None and self.split_bf16 and self._terms[1] == TERMS_F16X2
                    and _use_spw(N, Ho, Wo, conv.out_channels, ins[0].C)):
                # decoder conv1 with many up-sampled channels: their filter gradient as nine plain GEMMs between the low-resolution source
                # and tap-aligned 2x2 box sums of dy (conv_spw.hip); the skip channels' columns from the 3x3 kernel on the skip source alone
                cu_ = ins[0].C
                wu = sc_wgrad_args()
                wu.dy, wu.nsrc = dy, 1
                wu.src[0] = wa.src[0]
                wu.N, wu.H, wu.W, wu.Cout, wu.Cin, wu.ks = N, Ho, Wo, o.C, conv.in_channels, 3
                wu.part, wu.part_floats, wu.dw = None, 0, gv(conv.weight).data_ptr()
                wu.terms, wu.absmax = self._terms[1], gmax_slot.get(o.name)
                wu.xbound[0] = self._xbound(plan, ins[0])
                wsb = C.c_void_p((plan.spw_ws.data_ptr() + 255) & ~255)
                nb = plan.spw_ws.numel() - 256
                tok = self._pb("k_wgrad3_bx3 (+reduce)", flop, 0.0, 2.0 * N * (Ho // 2) * (Wo // 2) * 9 * conv.out_channels * cu_
                               + 2.0 * N * Ho * Wo * 9 * conv.out_channels * (conv.in_channels - cu_))
                wgrad_launch(lambda sx, wu=wu, wsb=wsb, nb=nb: check(lib.sc_conv3x3_sp_wgrad(C.byref(wu), wsb, nb, sx)))
                if len(ins) == 2:
                    csk_ = ins[1].C
                    wk = sc_wgrad_args()
                    wk.dy, wk.nsrc = dy, 1
                    wk.src[0] = wa.src[1]
                    wk.N, wk.H, wk.W, wk.Cout, wk.Cin, wk.ks = N, Ho, Wo, o.C, csk_, 3
                    wk.part, wk.part_floats, wk.dw = plan.ws.data_ptr(), plan.ws_floats, plan.spw_skip.data_ptr()
                    wk.terms, wk.absmax = self._terms[1], gmax_slot.get(o.name)
                    wk.xbound[0] = self._xbound(plan, ins[1])
                    wsk = lib.sc_conv3x3_wgrad_bx3 if (o.C >= 32 and csk_ >= 32) else lib.sc_conv2d_wgrad_mfma

                    def skip_cols(sx, wk=wk, wsk=wsk, csk_=csk_, cu_=cu_, conv=conv, o=o):
                        check(wsk(C.byref(wk), sx))
                        check(lib.sc_wgrad_scatter_cols(ptr(plan.spw_skip), ptr(gv(conv.weight)), o.C, csk_, conv.in_channels, cu_, sx))
                    wgrad_launch(skip_cols)
                self._pe(tok)
                wfn = None
            else:
                wfn = (lib.sc_conv3x3_wgrad_bx3 if (self.split_bf16 and ks == 3 and conv.out_channels >= 32 and conv.in_channels >= 32)
                       else lib.sc_conv2d_wgrad_mfma)     # 16-channel layers outside the cases below stay on the fp32 MFMA
            if (wfn is not None and self.split_bf16 and self.thin16 and ks == 3 and self._terms[1] == TERMS_F16X2 and len(ins) == 1
                    and conv.out_channels <= 16 and conv.in_channels in (16, 32) and Wo % 2 == 0):
                wfn = lib.sc_conv3x3_wgrad_thin16     # decoder.blocks.4: two fp16 terms on the 16x16x32 MFMA (was MFMA-bound in fp32)
            tok = None if wfn is None else self._pb(
                "k_wgrad3_bx3 (+reduce)" if wfn is lib.sc_conv3x3_wgrad_bx3 else
                "k_wgrad_thin_h (+reduce)" if wfn is lib.sc_conv3x3_wgrad_thin16 else f"k_wgrad_mfma<{ks}> (+reduce)", flop)
            if wfn is None:
                pass                      # (the box-sum GEMM path above has queued this layer's weight gradient)
            elif ty == "pw" and i in plan.pw_part:
                wa.part = plan.pw_part[i].data_ptr(); wa.part_floats = plan.pw_part[i].numel()
                pend = sc_wgrad_pending()
                wdef = (lib.sc_conv1x1_wgrad_pw3 if _use_pw3(2, N, Ho * Wo, conv.in_channels, conv.out_channels)
                        else lib.sc_conv2d_wgrad_mfma_deferred)
                wgrad_launch(lambda sx, wa=wa, pend=pend, wdef=wdef: check(wdef(C.byref(wa), C.byref(pend), sx)), second=True)
                pw_pending.append(pend)
                if i == last_pw:
                    # every pointwise layer has been walked: queue the batched reduction NOW, behind this layer's weight gradient, where it
                    # runs beside the main stream's last data-gradient kernels (features.1's depthwise backward, 0.16 ms).  At the end of
                    # the walk it was the side stream's last launch, 67 us exposed behind the stem's weight gradient before Adam.
                    reduce_pointwise_batch()
            else:
                wgrad_launch(lambda sx, wfn=wfn, wa=wa: check(wfn(C.byref(wa), sx)))
            self._pe(tok)
            # data gradient
            if ins[0].kind == "input":
                continue
            ent = self._wpk[i]
            a = sc_conv_args()
            a.nsrc = 1
            a.src[0] = dy
            a.wpk = ent["b"].data_ptr()
            a.N, a.H, a.W, a.Cout = N, Ho, Wo, conv.in_channels
            a.ks, a.co_t = ks, ent["cot_b"]
            a.terms = ent["terms_b"]
            a.absmax = gmax_slot.get(o.name)
            if ent["bx3_b"]:
                conv_dgrad = lib.sc_conv3x3_bx3
            elif ty == "pw" and _use_pw3(1, N, Ho * Wo, conv.in_channels, conv.out_channels):
                conv_dgrad = lib.sc_conv1x1_pw3
                a.wpk = ent["pb"].data_ptr()
            elif _use_ksplit(N, Ho * Wo, conv.out_channels, conv.in_channels, ks):
                conv_dgrad = lib.sc_conv1x1_ksplit
            else:
                conv_dgrad = lib.sc_conv2d_mfma
            a.add0 = None; a.add1 = None; a.stats = None
            a.accum0 = a.accum1 = 0
            # (sc_bnr_args are taken by k_conv3_bx3, not by its wave-specialised variant, which sc_conv3x3_bx3 picks from 16 K chunks up)
            bx3_plain = ent["bx3_b"] and not (a.terms == TERMS_F16X2 and (conv.out_channels + 15) // 16 >= 16 and conv.out_channels <= 256)
            # algorithmic bytes of the data gradient: g and y of the output once each, the input gradient once, the filter
            gin_elems = N * conv.in_channels * Ho * Wo
            if op.get("up") and ent["bx3_b"]:      # the upsampled source's gradient is stored 2x2-summed (quarter size)
                gin_elems -= N * ins[0].C * Ho * Wo * 3 // 4
            thin_b = (ent["tb"] is not None and not op.get("up") and ins[0].name not in written and res_of.get(ins[0].name) is None)
            # decoder.blocks.4.conv1: 16 gradient channels -> 32 channels at half resolution, the sub-pixel form on the thin layer's MFMA
            # (sc_conv3x3_thin16 with down0: 251 -> see DESIGN 16; STARCOP_THIN_SPD=0: sc_conv3x3_bx3 with its summing store)
            thin_sd = (ent.get("tsd") is not None and op.get("up") and len(ins) == 1 and a.terms == TERMS_F16X2 and _THIN_SPD and not _BNR
                       and Ho % 2 == 0 and Wo % 2 == 0)
            fle = None
            if (op.get("up") and ent.get("spd") is not None and ent["terms_b"] in _SP_TERMS and self.split_bf16
                    and _use_spd(N, Ho, Wo, ins[0].C, conv.in_channels - ins[0].C)):
                cu_ = ins[0].C          # up-sampled channels: 4 parity planes x 4 taps per low-resolution pixel; skip channels: the 3x3 form
                fle = (2.0 * N * (Ho // 2) * (Wo // 2) * 16 * conv.out_channels * (-(-cu_ // 128) * 128 + (128 * -(-(conv.in_channels - cu_) // 32) if ent["spd_stiles"] else 0))
                       + (0.0 if (ent["spd_vskip"] or ent["spd_stiles"]) else 2.0 * N * Ho * Wo * 9 * conv.out_channels * (conv.in_channels - cu_)))
            tok = self._pb("k_conv3_thin_h (fwd+dgrad)" if (thin_b or thin_sd) else
                           self._bx3_family("dgrad") if ent["bx3_b"] else f"k_conv_mfma<{ks}> (fwd+dgrad)", flop,
                           4.0 * (2 * N * o.C * Ho * Wo + gin_elems + conv.weight.numel()), fle)
            if (op.get("up") and ent.get("spd") is not None and ent["terms_b"] in _SP_TERMS and self.split_bf16
                    and _use_spd(N, Ho, Wo, ins[0].C, conv.in_channels - ins[0].C)):
                if ent["spd_vskip"] or ent["spd_stiles"]:
                    # <= 64 up-sampled + <= 16 skip channels (decoder.blocks.3): the skip channels' gradient as virtual channels of the
                    # 128-channel tile's second half -- dy (g, y) is staged ONCE for both gradients (289 us in two 3x3 launches before);
                    # skip tiles: as additional channel tiles of the launch (32 skip channels x 4 output parities each)
                    t_up, t_sk = ins
                    a.Cout, a.csplit = conv.in_channels, t_up.C
                    a.wpk = ent["spd"].data_ptr()
                    a.out0, a.out1 = plan.grad[t_up.name].data_ptr(), plan.grad[t_sk.name].data_ptr()
                    a.accum0, a.accum1, a.down0 = (1 if t_up.name in written else 0), (1 if t_sk.name in written else 0), 0
                    check(lib.sc_conv3x3_sp_dgrad(C.byref(a), st))
                    written.add(t_up.name); written.add(t_sk.name)
                    self._pe(tok)
                    continue
                # sub-pixel form: the up-sampled channels' gradient at half resolution from the four parity planes of dy (2.25x fewer
                # MFMAs, 128 output channels per staged patch), the skip channels' by the 3x3 kernel on their own 32-wide tiles
                t_up = ins[0]
                a.Cout = a.csplit = t_up.C
                a.wpk = ent["spd"].data_ptr()
                a.out0, a.out1 = plan.grad[t_up.name].data_ptr(), None
                a.accum0, a.down0 = (1 if t_up.name in written else 0), 0
                check(lib.sc_conv3x3_sp_dgrad(C.byref(a), st))
                written.add(t_up.name)
                if len(ins) == 2:
                    t_sk = ins[1]
                    a.Cout = a.csplit = t_sk.C
                    a.wpk, a.co_t = ent["bB"].data_ptr() + 4 * ent["bB_off"], 32
                    a.out0 = plan.grad[t_sk.name].data_ptr()
                    a.accum0 = 1 if t_sk.name in written else 0
                    check(lib.sc_conv3x3_bx3(C.byref(a), st))
                    written.add(t_sk.name)
                self._pe(tok)
                continue
            if thin_sd:
                t_up = ins[0]
                a.Cout = a.csplit = t_up.C
                a.wpk, a.co_t = ent["tsd"].data_ptr(), 16
                a.out0, a.out1 = plan.grad[t_up.name].data_ptr(), None
                a.accum0, a.down0, a.bnr = (1 if t_up.name in written else 0), 1, None
                check(lib.sc_conv3x3_thin16(C.byref(a), st))
                self._pe(tok)
                written.add(t_up.name)
                continue
            if op.get("up"):
                t_up = ins[0]
                fused_down = conv_dgrad is lib.sc_conv3x3_bx3      # the split-bf16 kernel stores the 2x2 sums itself
                if fused_down and ent.get("bA") is not None and self.split_dgrad_launch:
                    t_sk = ins[1]
                    a.Cout = a.csplit = t_up.C
                    a.wpk, a.co_t = ent["bA"].data_ptr(), 64
                    a.out0, a.out1 = plan.grad[t_up.name].data_ptr(), None
                    a.accum0, a.down0 = (1 if t_up.name in written else 0), 1
                    b_ = bnr_for(t_up, lib.sc_stat_rows(STAT_CONV3, N, Ho, Wo)) if bx3_plain else None
                    a.bnr = C.addressof(b_) if b_ is not None else None
                    check(conv_dgrad(C.byref(a), st))
                    a.bnr = None
                    a.Cout = a.csplit = t_sk.C
                    a.wpk, a.co_t = ent["bB"].data_ptr() + 4 * ent["bB_off"], 32
                    a.out0 = plan.grad[t_sk.name].data_ptr()
                    a.accum0, a.down0 = (1 if t_sk.name in written else 0), 0
                    check(conv_dgrad(C.byref(a), st))
                    self._pe(tok)
                    written.add(t_sk.name); written.add(t_up.name)
                    continue
                a.csplit = t_up.C
                if fused_down:
                    a.out0 = plan.grad[t_up.name].data_ptr()
                    a.accum0 = 1 if t_up.name in written else 0
                    a.down0 = 1
                else:
                    a.out0 = plan.up_tmp.data_ptr()
                if len(ins) == 2:
                    t_sk = ins[1]
                    a.out1 = plan.grad[t_sk.name].data_ptr()
                    a.accum1 = 1 if t_sk.name in written else 0
                    written.add(t_sk.name)
                else:
                    a.out1 = None
                b_ = None
                if fused_down and bx3_plain and (a.csplit == a.Cout or a.csplit % a.co_t == 0):
                    b_ = bnr_for(t_up, lib.sc_stat_rows(STAT_CONV3, N, Ho, Wo))
                a.bnr = C.addressof(b_) if b_ is not None else None
                check(conv_dgrad(C.byref(a), st))
                self._pe(tok)
                if not fused_down:
                    check(lib.sc_downsum2x2(ptr(plan.up_tmp), ptr(plan.grad[t_up.name]), 1 if t_up.name in written else 0,
                                            N, t_up.C, Ho // 2, Wo // 2, st))
                written.add(t_up.name)
            else:
                tin = ins[0]
                a.out0 = plan.grad[tin.name].data_ptr(); a.out1 = None
                a.csplit = conv.in_channels
                a.accum0 = 1 if tin.name in written else 0
                z = res_of.get(tin.name)
                if z is not None:
                    a.add0 = plan.grad[z].data_ptr()
                if thin_b:
                    conv_dgrad = lib.sc_conv3x3_thin16
                    a.wpk = ent["tb"].data_ptr()
                b_ = None
                if ks == 3 and (thin_b or (conv_dgrad is lib.sc_conv3x3_bx3 and bx3_plain)):
                    b_ = bnr_for(tin, lib.sc_stat_rows(STAT_CONV3, N, Ho, Wo), enabled=_BNR or (thin_b and _BNR_THIN))
                elif (conv_dgrad is lib.sc_conv1x1_pw3 and self.pw_bnr and z is None and not a.accum0
                      and N * Ho * Wo <= self.bn_small_max and tin.C >= 64):
                    # a projection's data gradient at <= 32 x 32 (the register-only pointwise kernel): it writes the COMPLETE gradient of the
                    # depthwise output d, so its epilogue leaves d's BatchNorm-backward sums (one row per 32-pixel block) and the dependent
                    # sc_bn_bwd_small launch over (gradient, d) -- 8 us alone, ~21 us under the weight-gradient stream -- becomes a finalize
                    b_ = bnr_for(tin, -(-(N * Ho * Wo) // 32), enabled=True)
                a.bnr = C.addressof(b_) if b_ is not None else None
                check(conv_dgrad(C.byref(a), st))
                self._pe(tok)
                written.add(tin.name)
        reduce_pointwise_batch()
        cast_dw_gradients()
        if side is not None:
            # join: every weight gradient is in the flat buffer before Adam / all-reduce.  With a gradient exchange to follow
            # (on_tail_ready: the data-parallel path) a default event, otherwise the device-scope one
            if exchange_follows if exchange_follows is not None else on_tail_ready is not None:
                main.wait_stream(side)
                if side2 is not None:
                    main.wait_stream(side2)
            else:
                wait_stream(main, side)
                if side2 is not None:
                    wait_stream(main, side2)

    # ------------------------------------------------------------------------------------------
    def forward(self, x, normalizer_consts=None):
        """(B,C,H,W) -> (B,1,H,W) logits.  ``normalizer_consts`` (C,8) fuses DataNormalizer.normalize_x into the stem."""
        need_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if need_grad:
            self._ensure_flat()
            return _UNetFunction.apply(self, x, normalizer_consts, *list(self.parameters()))
        plan = self._forward_impl(x, normalizer_consts, self.training, False)
        return plan.buf["logits"].clone()


class _UNetFunction(torch.autograd.Function):
    """Whole-network autograd node: backward runs the HIP backward pass into the flat grad buffer."""

    @staticmethod
    def forward(ctx, net, x, x_cst, *params):
        plan = net._forward_impl(x, x_cst, True, True)
        ctx.net, ctx.plan, ctx.generation = net, plan, plan.generation
        return plan.buf["logits"].clone()

    @staticmethod
    def backward(ctx, g):
        net, plan = ctx.net, ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("HyperStarcopUNet.backward: another forward of the same input shape ran after this one; the "
                               "network keeps ONE set of activation buffers per (N, H, W), so call backward() before the next "
                               "train-mode forward of that shape (or use a different batch shape)")
        params = list(net.parameters())
        # Gradient accumulation (accumulate_grad_batches > 1, zero_grad(set_to_none=False)): a p.grad left over from the last
        # backward IS a view of the flat gradient buffer, which _backward_impl overwrites.  Move those accumulators onto a
        # snapshot first and hand autograd fresh copies, so AccumulateGrad computes old + new (never new + new).
        aliased = [p.grad is not None and p.grad.data_ptr() == net._grad_view(p).data_ptr() for p in params]
        if any(aliased):
            snap = net._gflat.clone()
            base = net._gflat.data_ptr()
            for p, al in zip(params, aliased):
                if al:
                    off = (p.grad.data_ptr() - base) // 4
                    p.grad = snap[off:off + p.numel()].view(p.shape)
        net._backward_impl(plan, g.float())
        grads = [net._grad_view(p).clone() if al else net._grad_view(p) for p, al in zip(params, aliased)]
        return (None, None, None) + tuple(grads)
