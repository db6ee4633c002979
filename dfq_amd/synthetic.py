"""Synthetic networks for the BASELINE.json configurations (bench and test infrastructure).

There is no network access for checkpoints, so every configuration is a random-init network of
the named architecture with randomised BatchNorm statistics.  Only the layer
shapes and the graph topology matter to the calibration passes.

  * ``mobilenet_v2``    -- 52 Conv2d + 1 Linear = 53 layers, 3 469 760 weights (configs 1, 2, 5);
                            same block table as modeling/classification/MobileNetV2.py:62-98.
  * ``resnet18``        -- 20 Conv2d + 1 Linear, BasicBlock residual adds (config 3).
  * ``deeplab_mnv2``    -- DeepLab-v3+ with a MobileNetV2 backbone, ASPP ``torch.cat`` and decoder
                            (config 4; topology of modeling/segmentation/deeplab.py:9-34).
  * ``tiny_*``          -- few-layer nets that exercise every pairing case in seconds.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fxgraph import trace


# ------------------------------------------------------------------------------------------
# MobileNetV2
# ------------------------------------------------------------------------------------------
def _conv_bn_relu(inp, oup, k, stride, pad, groups=1, relu=True, dilation=1):
    layers = [nn.Conv2d(inp, oup, k, stride, pad, dilation=dilation, groups=groups, bias=False),
              nn.BatchNorm2d(oup)]
    if relu:
        layers.append(nn.ReLU6(inplace=True))
    return layers


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio, dilation=1):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers += _conv_bn_relu(inp, hidden, 1, 1, 0)
        layers += _conv_bn_relu(hidden, hidden, 3, stride, dilation, groups=hidden, dilation=dilation)
        layers += _conv_bn_relu(hidden, oup, 1, 1, 0, relu=False)
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        if self.use_res_connect:
            return x + self.conv(x)
        return self.conv(x)


_MNV2_TABLE = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
               (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


class MobileNetV2(nn.Module):
    def __init__(self, n_class=1000, width_mult=1.0, last_channel=1280):
        super().__init__()
        c_in = int(32 * width_mult)
        feats = [nn.Sequential(*_conv_bn_relu(3, c_in, 3, 2, 1))]
        for t, c, n, s in _MNV2_TABLE:
            c_out = int(c * width_mult)
            for i in range(n):
                feats.append(InvertedResidual(c_in, c_out, s if i == 0 else 1, t))
                c_in = c_out
        self.last_channel = int(last_channel * max(1.0, width_mult)) if width_mult >= 1.0 \
            else int(last_channel * width_mult)
        feats.append(nn.Sequential(*_conv_bn_relu(c_in, self.last_channel, 1, 1, 0)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Linear(self.last_channel, n_class)

    def forward(self, x):
        x = self.features(x)
        x = torch.mean(x.view(x.size(0), x.size(1), -1), -1)
        return self.classifier(x)


# ------------------------------------------------------------------------------------------
# ResNet-18
# ------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, inp, oup, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, oup, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(oup)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(oup, oup, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(oup)
        self.downsample = None
        if stride != 1 or inp != oup:
            self.downsample = nn.Sequential(nn.Conv2d(inp, oup, 1, stride, bias=False),
                                            nn.BatchNorm2d(oup))
        self.relu2 = nn.ReLU(inplace=True)

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        out = out + idt
        return self.relu2(out)


class ResNet18(nn.Module):
    def __init__(self, n_class=1000, widths=(64, 128, 256, 512)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, widths[0], 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(widths[0])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        blocks = []
        c_in = widths[0]
        for i, c in enumerate(widths):
            blocks.append(BasicBlock(c_in, c, 1 if i == 0 else 2))
            blocks.append(BasicBlock(c, c, 1))
            c_in = c
        self.layers = nn.Sequential(*blocks)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(c_in, n_class)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layers(x)
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


# ------------------------------------------------------------------------------------------
# DeepLab-v3+ (MobileNetV2 backbone)
# ------------------------------------------------------------------------------------------
class _PadConvBlock(nn.Module):
    """MobileNetV2 block of the segmentation backbone (modeling/segmentation/backbone/mobilenet.py:25-67): the explicit
    F.pad sits at the block's INPUT -- in front of the 1x1 expansion when there is one -- the depthwise conv has no padding
    of its own, and the residual is ``x + conv(pad(x))`` (images/graph_deeplab.png: BatchNorm2d_9 -> F.pad_10 -> Conv2d_11)."""

    def __init__(self, inp, oup, stride, dilation, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        self.pad = dilation                      # fixed_padding of a 3x3 kernel: (3 + 2 (d - 1) - 1) // 2 = d either side
        layers = []
        if expand_ratio != 1:
            layers += _conv_bn_relu(inp, hidden, 1, 1, 0)
        layers += _conv_bn_relu(hidden, hidden, 3, stride, 0, groups=hidden, dilation=dilation)
        layers += _conv_bn_relu(hidden, oup, 1, 1, 0, relu=False)
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        x_pad = F.pad(x, (self.pad, self.pad, self.pad, self.pad))
        if self.use_res_connect:
            return x + self.conv(x_pad)
        return self.conv(x_pad)


class _ASPP(nn.Module):
    def __init__(self, inp=320, mid=256, dilations=(1, 6, 12, 18)):
        super().__init__()
        self.b0 = nn.Sequential(nn.Conv2d(inp, mid, 1, bias=False), nn.BatchNorm2d(mid), nn.ReLU())
        self.b1 = nn.Sequential(nn.Conv2d(inp, mid, 3, padding=dilations[1], dilation=dilations[1],
                                          bias=False), nn.BatchNorm2d(mid), nn.ReLU())
        self.b2 = nn.Sequential(nn.Conv2d(inp, mid, 3, padding=dilations[2], dilation=dilations[2],
                                          bias=False), nn.BatchNorm2d(mid), nn.ReLU())
        self.b3 = nn.Sequential(nn.Conv2d(inp, mid, 3, padding=dilations[3], dilation=dilations[3],
                                          bias=False), nn.BatchNorm2d(mid), nn.ReLU())
        self.gp = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(inp, mid, 1, bias=False),
                                nn.BatchNorm2d(mid), nn.ReLU())
        self.proj = nn.Sequential(nn.Conv2d(mid * 5, mid, 1, bias=False), nn.BatchNorm2d(mid),
                                  nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x):
        x0, x1, x2, x3 = self.b0(x), self.b1(x), self.b2(x), self.b3(x)
        x4 = F.interpolate(self.gp(x), size=(x3.shape[2], x3.shape[3]), mode='bilinear',
                           align_corners=True)
        return self.proj(torch.cat((x0, x1, x2, x3, x4), dim=1))


class _Decoder(nn.Module):
    def __init__(self, n_class=21, low=24, mid=256):
        super().__init__()
        self.low = nn.Sequential(nn.Conv2d(low, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU())
        self.last = nn.Sequential(nn.Conv2d(mid + 48, mid, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(mid), nn.ReLU(), nn.Dropout(0.5),
                                  nn.Conv2d(mid, mid, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(mid), nn.ReLU(), nn.Dropout(0.1),
                                  nn.Conv2d(mid, n_class, 1))

    def forward(self, x, low):
        low = self.low(low)
        x = F.interpolate(x, size=(low.shape[2], low.shape[3]), mode='bilinear', align_corners=True)
        return self.last(torch.cat((x, low), dim=1))


class DeepLabMNV2(nn.Module):
    """61 convs, 5 780 288 weights -- the count SURVEY.md 8 quotes for the reference's DeepLab."""

    def __init__(self, n_class=21, output_stride=16):
        super().__init__()
        feats = [nn.Sequential(*_conv_bn_relu(3, 32, 3, 2, 1))]
        c_in, cur_stride, rate = 32, 2, 1
        for t, c, n, s in _MNV2_TABLE:
            if cur_stride == output_stride:
                stride, dilation = 1, rate
                rate *= s
            else:
                stride, dilation = s, 1
                cur_stride *= s
            for i in range(n):
                feats.append(_PadConvBlock(c_in, c, stride if i == 0 else 1, dilation, t))
                c_in = c
        self.low_level = nn.Sequential(*feats[:4])
        self.high_level = nn.Sequential(*feats[4:])
        self.aspp = _ASPP(320, 256)
        self.decoder = _Decoder(n_class, 24, 256)

    def forward(self, inp):
        low = self.low_level(inp)
        x = self.high_level(low)
        x = self.decoder(self.aspp(x), low)
        return F.interpolate(x, size=(inp.shape[2], inp.shape[3]), mode='bilinear', align_corners=True)


# ------------------------------------------------------------------------------------------
# tiny nets (fast parity cases)
# ------------------------------------------------------------------------------------------
class TinyMobile(nn.Module):
    """stem 3x3 -> [dw, pw] -> inverted residual (with add) -> inverted residual -> 1x1 -> mean -> fc."""

    def __init__(self, n_class=10):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_relu(3, 8, 3, 2, 1))
        self.b0 = InvertedResidual(8, 8, 1, 1)
        self.b1 = InvertedResidual(8, 12, 2, 6)
        self.b2 = InvertedResidual(12, 12, 1, 6)
        self.head = nn.Sequential(*_conv_bn_relu(12, 40, 1, 1, 0))
        self.fc = nn.Linear(40, n_class)

    def forward(self, x):
        x = self.head(self.b2(self.b1(self.b0(self.stem(x)))))
        x = torch.mean(x.view(x.size(0), x.size(1), -1), -1)
        return self.fc(x)


class TinyTail(nn.Module):
    """MobileNetV2's tail in miniature -- depthwise -> 1x1 -> 1x1 -> mean -> fc, one chain of five paired layers -- with a 1x1
    layer of 200 x 48 = 9600 weights in its middle: more than one LDS tile of the resident equalisation engine (8192 floats), so
    its column statistics are merged over several row blocks (the slot all-reduce of dfq_le_resident.hip) even on the CPU emulation."""

    def __init__(self, n_class=10):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_relu(3, 16, 3, 2, 1))
        self.dw = nn.Sequential(*_conv_bn_relu(16, 16, 3, 1, 1, groups=16))
        self.pw1 = nn.Sequential(*_conv_bn_relu(16, 48, 1, 1, 0))
        self.pw2 = nn.Sequential(*_conv_bn_relu(48, 200, 1, 1, 0))
        self.fc = nn.Linear(200, n_class)

    def forward(self, x):
        x = self.pw2(self.pw1(self.dw(self.stem(x))))
        x = torch.mean(x.view(x.size(0), x.size(1), -1), -1)
        return self.fc(x)


class TinyRes(nn.Module):
    def __init__(self, n_class=7):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.l1 = BasicBlock(8, 8, 1)
        self.l2 = BasicBlock(8, 16, 2)
        self.l3 = BasicBlock(16, 16, 1)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(16, n_class)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.l3(self.l2(self.l1(x)))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class TinyCat(nn.Module):
    """Two conv branches concatenated, grouped conv, conv with a real bias."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 8, 3, 1, 1, bias=True), nn.BatchNorm2d(8), nn.ReLU())
        self.a = nn.Sequential(nn.Conv2d(8, 6, 1, bias=False), nn.BatchNorm2d(6), nn.ReLU())
        self.b = nn.Sequential(nn.Conv2d(8, 10, 3, padding=2, dilation=2, bias=False),
                               nn.BatchNorm2d(10), nn.ReLU())
        self.proj = nn.Sequential(nn.Conv2d(16, 16, 1, bias=False), nn.BatchNorm2d(16), nn.ReLU())
        self.grp = nn.Sequential(nn.Conv2d(16, 32, 3, padding=1, groups=4, bias=False),
                                 nn.BatchNorm2d(32), nn.ReLU())
        self.last = nn.Conv2d(32, 5, 1)

    def forward(self, x):
        x = self.stem(x)
        x = torch.cat((self.a(x), self.b(x)), dim=1)
        return self.last(self.grp(self.proj(x)))


class TinyWide(nn.Module):
    """Geometry corner cases of the kernels: 3-input stem, grouped 3x3 (4 groups), depthwise 5x5, a 1x1
    expansion to 1600 channels (a row of > 1536 inputs for the classifier) and a 70-input 1x1."""

    def __init__(self, n_class=9):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_relu(3, 40, 3, 2, 1))
        self.grp = nn.Sequential(*_conv_bn_relu(40, 80, 3, 1, 1, groups=4))
        self.dw = nn.Sequential(*_conv_bn_relu(80, 80, 5, 1, 2, groups=80))
        self.pw = nn.Sequential(*_conv_bn_relu(80, 70, 1, 1, 0))
        self.head = nn.Sequential(*_conv_bn_relu(70, 1600, 1, 1, 0))
        self.fc = nn.Linear(1600, n_class)

    def forward(self, x):
        x = self.head(self.pw(self.dw(self.grp(self.stem(x)))))
        x = torch.mean(x.view(x.size(0), x.size(1), -1), -1)
        return self.fc(x)


class TinyHead(nn.Module):
    """Layers that are fed by a conv / linear WITHOUT batch norm (case d of set_quant_minmax,
    layer_transform.py:451-466): a plain 3x3 conv, a plain grouped conv and a plain linear layer sit
    between a BN and the next layer."""

    def __init__(self, n_class=5):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_relu(3, 8, 3, 2, 1))
        self.plain = nn.Conv2d(8, 12, 3, 1, 1, bias=True)
        self.mid = nn.Sequential(*_conv_bn_relu(12, 16, 1, 1, 0))
        self.plain_g = nn.Conv2d(16, 16, 3, 1, 1, groups=4, bias=True)
        self.tail = nn.Sequential(*_conv_bn_relu(16, 24, 1, 1, 0))
        self.fc1 = nn.Linear(24, 20)
        self.fc2 = nn.Linear(20, n_class)

    def forward(self, x):
        x = self.tail(self.plain_g(self.mid(self.plain(self.stem(x)))))
        x = torch.mean(x.view(x.size(0), x.size(1), -1), -1)
        return self.fc2(self.fc1(x))


class TinySeg(nn.Module):
    """DeepLab in miniature: every functional op the reference quantises on a segmentation graph
    (utils/layer_transform.py:10-14) -- F.interpolate behind a BN + ReLU (ASPP image pooling, decoder upsampling), behind a
    plain conv (the logits), torch.cat of branches, a residual add, and F.softmax on the upsampled logits."""

    def __init__(self, n_class=4):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_relu(3, 8, 3, 2, 1))
        self.low = InvertedResidual(8, 8, 1, 2)
        self.high = InvertedResidual(8, 12, 2, 3)
        self.b0 = nn.Sequential(nn.Conv2d(12, 6, 1, bias=False), nn.BatchNorm2d(6), nn.ReLU())
        self.b1 = nn.Sequential(nn.Conv2d(12, 6, 3, padding=2, dilation=2, bias=False), nn.BatchNorm2d(6), nn.ReLU())
        self.gp = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(12, 6, 1, bias=False), nn.BatchNorm2d(6), nn.ReLU())
        self.proj = nn.Sequential(nn.Conv2d(18, 10, 1, bias=False), nn.BatchNorm2d(10), nn.ReLU(), nn.Dropout(0.5))
        self.lowp = nn.Sequential(nn.Conv2d(8, 4, 1, bias=False), nn.BatchNorm2d(4), nn.ReLU())
        self.last = nn.Sequential(nn.Conv2d(14, 10, 3, padding=1, bias=False), nn.BatchNorm2d(10), nn.ReLU(),
                                  nn.Conv2d(10, n_class, 1))

    def forward(self, inp):
        low = self.low(self.stem(inp))
        x = self.high(low)
        x0, x1 = self.b0(x), self.b1(x)
        x2 = F.interpolate(self.gp(x), size=(x1.shape[2], x1.shape[3]), mode='bilinear', align_corners=True)
        x = self.proj(torch.cat((x0, x1, x2), dim=1))
        lowp = self.lowp(low)
        x = F.interpolate(x, size=(lowp.shape[2], lowp.shape[3]), mode='bilinear', align_corners=True)
        x = self.last(torch.cat((x, lowp), dim=1))
        x = F.interpolate(x, size=(inp.shape[2], inp.shape[3]), mode='bilinear', align_corners=True)
        return F.softmax(x, dim=1)


# ------------------------------------------------------------------------------------------
# factory helpers
# ------------------------------------------------------------------------------------------
def init_weights(model, gen):
    """Reference-style init (MobileNetV2.py:116-129): conv N(0, sqrt(2/(k*k*O))), fc N(0, 0.01),
    followed by randomised BN affine/statistics so that folding BN is non-trivial."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n), generator=gen)
            if m.bias is not None:
                m.bias.data.normal_(0, 0.1, generator=gen)
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.01, generator=gen)
            m.bias.data.zero_()
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            m.weight.data.copy_(torch.rand(c, generator=gen) * 1.5 + 0.5)          # U(0.5, 2)
            m.bias.data.copy_(torch.randn(c, generator=gen) * 0.5)                 # N(0, 0.5)
            m.running_mean.copy_(torch.randn(c, generator=gen) * 0.5)              # N(0, 0.5)
            m.running_var.copy_(torch.rand(c, generator=gen) * 1.9 + 0.1)          # U(0.1, 2)


def relu6_to_relu(model):
    """The ``--relu`` switch of main_cls.py:126-127 (ReLU6 -> ReLU) done in place."""
    for name, child in model.named_children():
        if isinstance(child, nn.ReLU6):
            setattr(model, name, nn.ReLU(inplace=False))
        else:
            relu6_to_relu(child)
    return model


_FACTORY = {
    'mobilenet_v2': MobileNetV2, 'resnet18': ResNet18, 'deeplab_mnv2': DeepLabMNV2,
    'tiny_mobile': TinyMobile, 'tiny_res': TinyRes, 'tiny_cat': TinyCat, 'tiny_wide': TinyWide, 'tiny_head': TinyHead, 'tiny_seg': TinySeg, 'tiny_tail': TinyTail,
}


def build(name, seed=0, keep_relu6=False, **kw):
    """Random-init ``name`` on CPU (deterministic for a given torch build), eval mode, ReLU6->ReLU
    (unless ``keep_relu6``: main_cls.py without ``--relu``).

    Returns (model, graph, bottoms) with the graph dicts in the reference's format.
    """
    gen = torch.Generator(device='cpu')
    gen.manual_seed(seed)
    with torch.no_grad():
        model = _FACTORY[name](**kw)
        init_weights(model, gen)
    model.eval()
    if not keep_relu6:
        relu6_to_relu(model)
    graph, bottoms = trace(model)
    return model, graph, bottoms
