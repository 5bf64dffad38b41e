"""ConvModule + BRICKS registry (/root/reference/python/jdet/models/utils/modules.py:44-221).

conv -> norm -> act bundle; ``bias='auto'`` means "bias iff no norm"; default
activation ReLU; kaiming init (callers such as S2ANetHead overwrite it)."""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.ops.bn_act import bias_act
from rs_detection_amd.ops.conv3x3 import (conv3x3_applies, conv3x3_same, fast_conv, conv3x3_mfma_applies,
                                          conv3x3_bias_relu)
from rs_detection_amd.utils.registry import BRICKS, build_from_cfg
from .weight_init import kaiming_init, constant_init

_FUSE_BIAS_RELU = True
_FUSE_BIAS_RELU_AMP = True      # also under bf16 autocast

BRICKS.register_module(name="Conv2d", module=nn.Conv2d)
BRICKS.register_module(name="ReLU", module=nn.ReLU)
BRICKS.register_module(name="LeakyReLU", module=nn.LeakyReLU)
BRICKS.register_module(name="GELU", module=nn.GELU)
BRICKS.register_module(name="Tanh", module=nn.Tanh)
BRICKS.register_module(name="Sigmoid", module=nn.Sigmoid)


@BRICKS.register_module(name="BN")
def _bn(in_channels, **kw):
    kw.pop("requires_grad", None)
    return nn.BatchNorm2d(in_channels, **kw)


@BRICKS.register_module(name="GN")
def _gn(num_channels, num_groups=32, **kw):
    kw.pop("requires_grad", None)
    return nn.GroupNorm(num_groups, num_channels, **kw)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), with_spectral_norm=False,
                 padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        assert act_cfg is None or isinstance(act_cfg, dict)
        assert padding_mode in ('zeros', 'circular'), "explicit padding layers are not on the S2ANet/ORCNN path"
        assert isinstance(order, tuple) and set(order) == {'conv', 'norm', 'act'}
        self.conv_cfg, self.norm_cfg, self.act_cfg, self.order = conv_cfg, norm_cfg, act_cfg, order
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')
        self.conv = build_from_cfg(dict(type='Conv2d') if conv_cfg is None else conv_cfg, BRICKS,
                                   in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                                   stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = self.conv.kernel_size, self.conv.stride, padding
        self.dilation, self.groups = self.conv.dilation, self.conv.groups
        if with_spectral_norm:
            self.conv = nn.utils.spectral_norm(self.conv)
        self.norm = "None"
        if self.with_norm:
            ch = out_channels if order.index('norm') > order.index('conv') else in_channels
            cfg = dict(norm_cfg)
            if cfg.get("type", "BN") == "GN":
                self.gn = build_from_cfg(cfg, BRICKS, num_channels=ch)
                self.norm = "gn"
            else:
                cfg.setdefault("type", "BN")
                self.bn = build_from_cfg(cfg, BRICKS, in_channels=ch)
                self.norm = "bn"
        if self.with_activation and act_cfg['type'] not in ['Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish']:
            cfg = dict(act_cfg)
            if cfg['type'] in ('ReLU', 'LeakyReLU'):
                cfg.setdefault('inplace', True)
            self.activate = build_from_cfg(cfg, BRICKS)
        self.init_weights()

    def init_weights(self):
        if not hasattr(self.conv, 'init_weights'):
            if self.with_activation and self.act_cfg['type'] == 'LeakyReLU':
                kaiming_init(self.conv, a=self.act_cfg.get('negative_slope', 0.01), nonlinearity='leaky_relu')
            else:
                kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(getattr(self, self.norm), 1, bias=0)

    def _fused_bias_relu(self, x, activate):
        """conv (bias) -> ReLU, the S2ANet head towers: run the convolution without its bias and apply bias + ReLU as one
        fused pass (ops/bn_act.py: bias_act).  Only for the plain case; anything else takes the generic loop."""
        conv = self.conv
        return (_FUSE_BIAS_RELU and activate and self.with_activation and not self.with_norm
                and self.order.index('conv') == 0
                and type(conv) is nn.Conv2d and conv.bias is not None and conv.padding_mode == 'zeros'
                and isinstance(getattr(self, 'activate', None), nn.ReLU) and x.is_cuda
                # under bf16 autocast only for a channels_last input (26.8 -> 26.3 ms/step); NCHW bf16 measured slower
                # (40.5 vs 32.9 ms/step: MIOpen's NCHW bf16 solvers prefer to own the bias)
                and ((x.dtype == torch.float32 and not torch.is_autocast_enabled())
                     or (_FUSE_BIAS_RELU_AMP and torch.is_autocast_enabled() and x.dim() == 4 and not x.is_contiguous()
                         and x.is_contiguous(memory_format=torch.channels_last))))

    def forward(self, x, activate=True, norm=True, canvas=None):
        """``canvas``: x is a pyramid canvas (ops/pyramid.CanvasLayout) -- the output's gap pixels are put back to zero
        so that the next convolution of the tower sees each level's zero padding."""
        if self._fused_bias_relu(x, activate):
            conv = self.conv
            if conv3x3_mfma_applies(x, conv):         # conv + bias + ReLU (+ gap mask) in one launch of our own kernel
                return conv3x3_bias_relu(x, conv, None if canvas is None else canvas.live)
            if conv3x3_applies(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
                y = conv3x3_same(x, conv.weight)      # backward-data through the forward solver (ops/conv3x3.py)
            else:
                y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if canvas is not None:
                from rs_detection_amd.ops.pyramid import canvas_bias_act
                return canvas_bias_act(y, conv.bias, canvas, relu=True)
            return bias_act(y, conv.bias, relu=True)
        if canvas is not None:
            out = self.forward(x, activate, norm)
            return out * canvas.live_f.to(out.dtype)
        for layer in self.order:
            if layer == 'conv':
                x = fast_conv(self.conv, x)     # the module itself unless a faster equivalent applies (ops/conv3x3.py)
            elif layer == 'norm' and norm and self.with_norm:
                x = getattr(self, self.norm)(x)
            elif layer == 'act' and activate and self.with_activation and hasattr(self, 'activate'):
                x = self.activate(x)
        return x
