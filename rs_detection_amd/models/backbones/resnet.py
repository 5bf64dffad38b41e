"""ResNet backbones (dense convolutions -> MIOpen / MFMA through torch).

Interface mirror of /root/reference/python/jdet/models/backbones/resnet.py:95-265:
``Resnet50(pretrained, frozen_stages, return_stages, norm_eval)``; ``train()``
re-freezes stem + the first ``frozen_stages`` layers and keeps EVERY BatchNorm in
eval mode when ``norm_eval`` (default True, :177-184).  Parameter names
(conv1/bn1/layerN.M.convK/bnK/downsample.0/1) match the reference state_dict.
``pretrained=True`` would fetch ``jittorhub://resnet50.pkl``; there is no network
here, so weights stay random-initialised (relu-invariant gaussian, fan_out).
"""
import torch
import torch.nn as nn

from rs_detection_amd.ops.bn_act import Forked, bn_act, bn_relu_maxpool
from rs_detection_amd.ops.conv1x1 import conv1x1
from rs_detection_amd.ops.bottleneck import bottleneck, bottleneck_applies
from rs_detection_amd.ops.conv_bn import conv_bn_act
from rs_detection_amd.ops.conv3x3 import fast_conv
from rs_detection_amd.utils.registry import BACKBONES

__all__ = ['ResNet', 'Resnet18', 'Resnet34', 'Resnet50', 'Resnet101', 'Resnet152']


def _conv(inp, out, k, stride=1, groups=1, dilation=1):
    pad = dilation if k == 3 else 0
    conv = nn.Conv2d(inp, out, k, stride=stride, padding=pad, groups=groups, bias=False, dilation=dilation)
    nn.init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu')
    return conv


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        if dilation > 1:
            raise NotImplementedError('Dilation > 1 not supported in BasicBlock')
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = bn_act(self.conv1(x), self.bn1)
        return bn_act(self.conv2(out), self.bn2, residual=idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1, self.bn1 = _conv(inplanes, width, 1), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = _conv(width, width, 3, stride, groups, dilation), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = _conv(width, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # bn -> (+ identity) -> relu as one pass each way when the BatchNorm is in eval mode (ops/bn_act.py)
        # conv1 / conv3 / a stride-1 downsample are 1x1: GEMMs on views when the step runs channels_last (ops/conv1x1.py)
        # in the bf16 channels_last step the 1x1 convolutions take their BatchNorm tail into the GEMM's epilogue: one
        # launch each (ops/conv_bn.py, csrc/gemm1x1_mfma.hip)
        # an identity block of the bf16 step as ONE autograd node with a hand-ordered backward (ops/bottleneck.py)
        # fp32 channels_last: the block's output goes on as a Forked pair (ops/bn_act.py) -- `a` to the next block's conv1,
        # `b` to its identity branch -- so that the two gradients are summed inside the BatchNorm backward kernel
        xa, xb = (x.a, x.b) if isinstance(x, Forked) else (x, x)
        if bottleneck_applies(self, xa):
            return bottleneck(self, xa)
        idt = xb if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], xb, relu=False)
        out = conv_bn_act(self.conv1, self.bn1, xa)
        out = bn_act(fast_conv(self.conv2, out), self.bn2)      # stride-1 3x3: backward-data via the forward solver
        return conv_bn_act(self.conv3, self.bn3, out, residual=idt, fork=True)


@BACKBONES.register_module()
class ResNet(nn.Module):
    def __init__(self, block, layers, return_stages=["layer4"], frozen_stages=-1, norm_eval=True, num_classes=None,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None):
        super().__init__()
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.inplanes, self.dilation = 64, 1
        rswd = replace_stride_with_dilation or [False, False, False]
        if len(rswd) != 3:
            raise ValueError('replace_stride_with_dilation should be None or a 3-element tuple, got {}'.format(rswd))
        self.groups, self.base_width = groups, width_per_group
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        nn.init.kaiming_normal_(self.conv1.weight, mode='fan_out', nonlinearity='relu')
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2, rswd[0])
        self.layer3 = self._make_layer(block, 256, layers[2], 2, rswd[1])
        self.layer4 = self._make_layer(block, 512, layers[3], 2, rswd[2])
        self.num_classes, self.return_stages = num_classes, return_stages
        if num_classes is not None:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)
        self._freeze_stages()

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down, self.groups, self.base_width, prev_dil)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation))
        return nn.Sequential(*layers)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad_(False)
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, 'layer{}'.format(i))
            m.eval()
            for p in m.parameters():
                p.requires_grad_(False)

    def set_channels_last(self, on=True):
        """Run the trunk in channels_last (NHWC): MIOpen's bf16 convolutions are NHWC-native and wrap every NCHW call
        in layout transposes (6.6 ms of a 28 ms bf16 step); the fused BatchNorm tails have NHWC kernels
        (ops/bn_act.py).  The input is converted once, the returned stage outputs go back to NCHW for the neck."""
        self.trunk_channels_last = bool(on)
        fmt = torch.channels_last if on else torch.contiguous_format
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=fmt)
        return self

    def forward(self, x):
        outs = []
        cl = getattr(self, "trunk_channels_last", False) and x.is_cuda
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        frozen_stem = self.frozen_stages >= 0
        with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen_stem):
            x = bn_relu_maxpool(self.conv1(x), self.bn1, self.maxpool)   # one pass when no gradient is recorded
        for i in range(1, 5):
            name = f"layer{i}"
            with torch.set_grad_enabled(torch.is_grad_enabled() and i > self.frozen_stages):
                x = getattr(self, name)(x)
            if name in self.return_stages:
                xo = x.a if isinstance(x, Forked) else x          # (the neck is a third reader: its gradient joins `a`'s)
                outs.append(xo.contiguous() if cl else xo)
        if isinstance(x, Forked):
            x = x.a
        if self.num_classes is not None:
            x = self.fc(torch.flatten(self.avgpool(x), 1))
            if "fc" in self.return_stages:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


def _factory(block, layers, hub):
    def make(pretrained=False, **kwargs):
        model = ResNet(block, layers, **kwargs)
        model.pretrained_source = f"jittorhub://{hub}.pkl" if pretrained else None
        from rs_detection_amd.runner.checkpoint import load_pretrained
        model.pretrained_report = load_pretrained(model, hub, pretrained)   # warns loudly when nothing is available
        return model
    return make


Resnet18 = BACKBONES.register_module(name="Resnet18", module=_factory(BasicBlock, [2, 2, 2, 2], "resnet18"))
Resnet34 = BACKBONES.register_module(name="Resnet34", module=_factory(BasicBlock, [3, 4, 6, 3], "resnet34"))
Resnet50 = BACKBONES.register_module(name="Resnet50", module=_factory(Bottleneck, [3, 4, 6, 3], "resnet50"))
Resnet101 = BACKBONES.register_module(name="Resnet101", module=_factory(Bottleneck, [3, 4, 23, 3], "resnet101"))
Resnet152 = BACKBONES.register_module(name="Resnet152", module=_factory(Bottleneck, [3, 8, 36, 3], "resnet152"))
