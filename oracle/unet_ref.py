"""CPU restatement (plain torch ops) of the network ``scripts/train.py`` trains.

TEST INFRASTRUCTURE -- not product code (see oracle/__init__.py).

Reference call site: ``starcop/models/model_module.py:244-251``

    smp.Unet(encoder_name='mobilenet_v2', encoder_weights=None,
             in_channels=C, classes=1, activation=None)

``segmentation_models_pytorch`` (requirements.txt:9, unpinned) and the
torchvision ``MobileNetV2`` it subclasses are NOT vendored under
/root/reference and are not installed, so this file restates their published
architecture (smp 0.3.x: ``encoders/mobilenet.py``, ``decoders/unet/decoder.py``,
``base/heads.py``; torchvision ``models/mobilenetv2.py``).  PARITY UNPINNED for
bytes; structure is pinned by

  * the parameter count the reference logged: 6 629 233 trainable for C=4
    (notebooks/(bonus)_training_demo.ipynb cell 19) -- asserted in
    tests/test_oracle.py,
  * ``state_dict`` key names (``encoder.features.N...``, ``decoder.blocks.N.convK.M``,
    ``segmentation_head.0``) so that reference checkpoints would load with
    ``strict=True``.

Every module below is an ordinary ``torch.nn`` module evaluated with stock
CPU kernels: this is the "reference CPU path" the HIP kernels are compared to.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision MobileNetV2 inverted-residual settings (t, c, n, s)
MBV2_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                 (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class ConvBNReLU6(nn.Sequential):
    """torchvision ``ConvBNReLU`` / ``Conv2dNormActivation``: [0]=conv, [1]=BN, [2]=ReLU6."""

    def __init__(self, cin, cout, k=3, stride=1, groups=1):
        super().__init__(
            nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout),
            nn.ReLU6(inplace=False),
        )


class InvertedResidual(nn.Module):
    """torchvision ``InvertedResidual``: keys ``conv.{0,1,2,3}``."""

    def __init__(self, cin, cout, stride, expand_ratio):
        super().__init__()
        hidden = int(round(cin * expand_ratio))
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU6(cin, hidden, k=1))
        layers += [
            ConvBNReLU6(hidden, hidden, k=3, stride=stride, groups=hidden),
            nn.Conv2d(hidden, cout, 1, 1, 0, bias=False),
            nn.BatchNorm2d(cout),
        ]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2Encoder(nn.Module):
    """smp ``MobileNetV2Encoder`` (classifier removed); skip taps after features 1,3,6,13,18."""

    def __init__(self, in_channels):
        super().__init__()
        feats = [ConvBNReLU6(in_channels, 32, k=3, stride=2)]
        cin = 32
        for t, c, n, s in MBV2_SETTINGS:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(ConvBNReLU6(cin, 1280, k=1))
        self.features = nn.Sequential(*feats)

    def forward(self, x):
        stages = (self.features[:2], self.features[2:4], self.features[4:7],
                  self.features[7:14], self.features[14:])
        outs = [x]
        for st in stages:
            x = st(x)
            outs.append(x)
        return outs


class Conv2dReLU(nn.Sequential):
    """smp ``Conv2dReLU(use_batchnorm=True)``: [0]=conv(bias=False), [1]=BN, [2]=ReLU."""

    def __init__(self, cin, cout):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=1, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU(inplace=False))


class DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = Conv2dReLU(cin + cskip, cout)
        self.conv2 = Conv2dReLU(cout, cout)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]           # 1280, 96, 32, 24, 16
        cins = [enc[0]] + list(DECODER_CHANNELS[:-1])    # 1280, 256, 128, 64, 32
        cskips = enc[1:] + [0]                           # 96, 32, 24, 16, 0
        self.blocks = nn.ModuleList(
            DecoderBlock(i, s, o) for i, s, o in zip(cins, cskips, DECODER_CHANNELS))

    def forward(self, feats):
        feats = feats[1:][::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UnetMobileNetV2(nn.Module):
    """``smp.Unet('mobilenet_v2', encoder_weights=None, in_channels=C, classes=K, activation=None)``."""

    def __init__(self, in_channels=4, classes=1):
        super().__init__()
        self.encoder = MobileNetV2Encoder(in_channels)
        self.decoder = UnetDecoder((in_channels, 16, 24, 32, 96, 1280))
        self.segmentation_head = nn.Sequential(
            nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        self.reset_parameters(in_channels)

    def reset_parameters(self, in_channels):
        # torchvision MobileNetV2.__init__ weight init
        for m in self.encoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if in_channels != 3:
            # smp patch_first_conv(pretrained=False): new weight + reset_parameters()
            self.encoder.features[0][0].reset_parameters()
        # smp initialize_decoder / initialize_head
        for m in self.decoder.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for m in self.segmentation_head.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        if x.shape[-1] % 32 or x.shape[-2] % 32:
            raise RuntimeError(f"Wrong input shape height={x.shape[-2]}, width={x.shape[-1]}: "
                               "must be divisible by 32")
        return self.segmentation_head(self.decoder(self.encoder(x)))


def layer_table(in_channels=4, hw=512):
    """(prefix, cin, cout, k, stride, groups, hin, hout) for every conv, in execution order."""
    rows = []
    h = hw
    rows.append(("encoder.features.0.0", in_channels, 32, 3, 2, 1, h, h // 2)); h //= 2
    cin, idx = 32, 1
    for t, c, n, s in MBV2_SETTINGS:
        for i in range(n):
            stride = s if i == 0 else 1
            hid = cin * t
            j = 0
            if t != 1:
                rows.append((f"encoder.features.{idx}.conv.0.0", cin, hid, 1, 1, 1, h, h)); j = 1
            rows.append((f"encoder.features.{idx}.conv.{j}.0", hid, hid, 3, stride, hid, h, h // stride))
            h //= stride
            rows.append((f"encoder.features.{idx}.conv.{j + 1}", hid, c, 1, 1, 1, h, h))
            cin = c; idx += 1
    rows.append(("encoder.features.18.0", cin, 1280, 1, 1, 1, h, h))
    cins = [1280 + 96, 256 + 32, 128 + 24, 64 + 16, 32]
    for b, (ci, co) in enumerate(zip(cins, DECODER_CHANNELS)):
        h *= 2
        rows.append((f"decoder.blocks.{b}.conv1.0", ci, co, 3, 1, 1, h, h))
        rows.append((f"decoder.blocks.{b}.conv2.0", co, co, 3, 1, 1, h, h))
    rows.append(("segmentation_head.0", 16, 1, 3, 1, 1, h, h))
    return rows


def conv_stack_ref(params, x, relu_last=True):
    """A bias-carrying stack ``conv(k, pad k//2) -> ReLU -> conv -> ReLU`` evaluated with the same stock CPU ops the
    restatement above uses (``F.conv2d`` / ``F.relu`` / autograd).  Restates the reference's in-repo blocks
    ``layer_factory.double_conv`` (/root/reference/starcop/models/architectures/layer_factory.py:4-9) and ``UNet.conv_last``
    (architectures/unet.py:21): the only convolution arithmetic the reference itself holds, and therefore what pins the
    convolution / ReLU forward and backward of this oracle to reference-executed numbers (golden G9,
    tests/golden/make_golden.py::g9_convblocks, checked in tests/test_oracle.py::test_g9_conv_blocks)."""
    for i, (w, b) in enumerate(params):
        x = F.conv2d(x, w, b, padding=w.shape[-1] // 2)
        if i + 1 < len(params) or relu_last:
            x = F.relu(x)
    return x


def simple_unet_ref(sd, x):
    """The reference's in-repo ``UNet`` (/root/reference/starcop/models/architectures/unet.py:7-51) as a function of its
    ``state_dict``: four double_conv stages with MaxPool2d(2) in between, three [bilinear x2 (align_corners=True) -> cat(skip) ->
    double_conv] stages, 1x1 head -- stock CPU ops.  Pinned by the reference's own forward of the full network (golden G9
    ``unet_full.*``, tests/test_oracle.py::test_g9_full_unet)."""
    def dc(name, t):
        return conv_stack_ref([(sd[f"{name}.0.weight"], sd[f"{name}.0.bias"]), (sd[f"{name}.2.weight"], sd[f"{name}.2.bias"])], t)
    c1 = dc("dconv_down1", x)
    c2 = dc("dconv_down2", F.max_pool2d(c1, 2))
    c3 = dc("dconv_down3", F.max_pool2d(c2, 2))
    t = dc("dconv_down4", F.max_pool2d(c3, 2))
    for name, skip in (("dconv_up3", c3), ("dconv_up2", c2), ("dconv_up1", c1)):
        t = dc(name, torch.cat([F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True), skip], dim=1))
    return F.conv2d(t, sd["conv_last.weight"], sd["conv_last.bias"])
