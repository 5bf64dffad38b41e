"""Visual Attention Network backbones (van_b0..b3) with the reference's module names / state_dict keys
(/root/reference/python/jdet/models/backbones/van.py:140-483): overlap patch embed (7x7 s4, then 3x3 s2) + BN,
blocks = BN -> LKA attention (1x1, GELU, depthwise 5x5, depthwise 7x7 dilation 3, 1x1, gate) and BN -> MLP
(1x1, depthwise 3x3, GELU, 1x1) with layer-scale, LayerNorm per stage.  Dense convs run in MIOpen via torch, the
depthwise ones in the LDS-tiled HIP stencil of csrc/dwconv.hip (ops/dwconv.py).
``pretrained=True`` would download ImageNet weights (no network here): weights stay random-initialised."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.ops import chan_layernorm, van_block, van_fused
from rs_detection_amd.ops.bn_act import scale_residual
from rs_detection_amd.ops.conv1x1 import conv1x1_nchw
from rs_detection_amd.ops.dwconv import DepthwiseConv2d
from rs_detection_amd.utils.registry import BACKBONES


def _init(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = DepthwiseConv2d(dim, 3, padding=1, bias=True)  # van.py:32; HIP stencil on the GPU (ops/dwconv.py)

    def forward(self, x, in_bias=None):
        return self.dwconv(x, in_bias)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Conv2d(in_features, hidden_features, 1)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Conv2d(hidden_features, out_features, 1)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        # fc1's bias is applied inside the depthwise kernel's load (ops/dwconv.py in_bias) instead of as a separate
        # pass over the hidden-width tensor (8x / 4x the block width); its gradient comes back from that kernel too
        return self.drop(self.fc2(self.hidden(x)))

    def hidden(self, x):
        """Everything up to fc2's input."""
        h = conv1x1_nchw(x, self.fc1.weight)
        return self.drop(self.act(self.dwconv(h, self.fc1.bias)))


class AttentionModule(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv0 = DepthwiseConv2d(dim, 5, padding=2)                         # van.py:56
        self.conv_spatial = DepthwiseConv2d(dim, 7, padding=9, dilation=3)      # van.py:57
        self.conv1 = nn.Conv2d(dim, dim, 1)

    def forward(self, x):
        # u * (conv1(...) + bias): the 1x1 convolution without its bias, bias + gate as one pass (ops/van_fused.py)
        a = self.conv_spatial(self.conv0(x))
        if van_fused.applies(x):
            return van_fused.gate(x, conv1x1_nchw(a, self.conv1.weight), self.conv1.bias)
        return x * self.conv1(a)


class SpatialAttention(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.proj_1 = nn.Conv2d(d_model, d_model, 1)
        self.activation = nn.GELU()
        self.spatial_gating_unit = AttentionModule(d_model)
        self.proj_2 = nn.Conv2d(d_model, d_model, 1)

    def forward(self, x):
        return self.proj_2(self.gated(x)) + x

    def gated(self, x):
        """Everything up to proj_2's input; proj_1's bias + GELU as one pass when the fused kernels apply."""
        if van_fused.applies(x) and isinstance(self.activation, nn.GELU) and self.activation.approximate == 'none':
            u = van_fused.bias_gelu(conv1x1_nchw(x, self.proj_1.weight), self.proj_1.bias)
        else:
            u = self.activation(self.proj_1(x))
        return self.spatial_gating_unit(u)


class DropPath(nn.Module):
    def __init__(self, p=0.):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0. or not self.training:
            return x
        keep = 1 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x / keep * mask


class Block(nn.Module):
    def __init__(self, dim, mlp_ratio=4., drop=0., drop_path=0., act_layer=nn.GELU):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(dim)
        self.attn = SpatialAttention(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.BatchNorm2d(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.layer_scale_1 = nn.Parameter(1e-2 * torch.ones(dim))
        self.layer_scale_2 = nn.Parameter(1e-2 * torch.ones(dim))

    def forward(self, x):
        # the whole block as ONE autograd node on our fp32 MFMA GEMMs with fused tails (ops/van_block.py) where it applies:
        # training-mode BatchNorms, idle drop-path / dropout, channel counts and map sizes the tiles divide
        if self._idle() and van_block.applies(self, x):
            return van_block.van_block(self, x)
        # x + drop_path(layer_scale * f) (van.py:121-122); the per-sample drop-path factor commutes with the per-channel
        # scale, so it is applied to f and scale + residual run as one fused pass (ops/bn_act.py: scale_residual)
        if self._fused(x):
            # the last 1x1 convolution of each half without its bias: bias (+ the attention's own shortcut) + layer scale
            # + residual as ONE pass each way (ops/van_fused.py); nothing sits between them when drop-path and dropout
            # are inactive
            xn = self.norm1(x)
            p = conv1x1_nchw(self.attn.gated(xn), self.attn.proj_2.weight)
            x = van_fused.residual(x, p, self.attn.proj_2.bias, xn, self.layer_scale_1)
            p = conv1x1_nchw(self.mlp.hidden(self.norm2(x)), self.mlp.fc2.weight)
            return van_fused.residual(x, p, self.mlp.fc2.bias, None, self.layer_scale_2)
        x = scale_residual(x, self.drop_path(self.attn(self.norm1(x))), self.layer_scale_1)
        return scale_residual(x, self.drop_path(self.mlp(self.norm2(x))), self.layer_scale_2)

    def _idle(self):
        idle_path = isinstance(self.drop_path, nn.Identity) or not self.training or self.drop_path.p == 0.
        idle_drop = not self.training or self.mlp.drop.p == 0.
        return idle_path and idle_drop

    def _fused(self, x):
        return van_fused.applies(x) and self._idle()


class OverlapPatchEmbed(nn.Module):
    def __init__(self, img_size=1024, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.BatchNorm2d(embed_dim)

    def forward(self, x):
        return self.norm(self.proj(x))


class VAN(nn.Module):
    def __init__(self, img_size=1024, in_chans=3, num_classes=10, embed_dims=[64, 128, 256, 512],
                 mlp_ratios=[4, 4, 4, 4], drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 depths=[3, 4, 6, 3], num_stages=4, flag=False, out_indices=(0, 1, 2)):
        super().__init__()
        self.depths, self.num_stages, self.out_indices = depths, num_stages, out_indices
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        cur = 0
        for i in range(num_stages):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(
                img_size=img_size if i == 0 else img_size // (2 ** (i + 1)), patch_size=7 if i == 0 else 3,
                stride=4 if i == 0 else 2, in_chans=in_chans if i == 0 else embed_dims[i - 1], embed_dim=embed_dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList([Block(dim=embed_dims[i], mlp_ratio=mlp_ratios[i],
                                                                drop=drop_rate, drop_path=dpr[cur + j])
                                                          for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
            cur += depths[i]
        self.apply(_init)

    def forward(self, x):
        B = x.shape[0]
        outs = []
        for i in range(self.num_stages):
            x = getattr(self, f"patch_embed{i + 1}")(x)
            _, _, H, W = x.shape
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x)
            norm = getattr(self, f"norm{i + 1}")
            if chan_layernorm.applies(x, norm):          # one launch on the NCHW map (csrc/van_ops.hip)
                x = chan_layernorm.chan_layer_norm(x, norm)
            else:
                x = norm(x.flatten(2).transpose(1, 2))
                x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
            if i in self.out_indices:
                outs.append(x)
        return outs


def _variant(dims, ratios, depths, name=None):
    def make(pretrained=False, **kwargs):
        m = VAN(embed_dims=dims, mlp_ratios=ratios, norm_layer=nn.LayerNorm, depths=depths, **kwargs)
        m.pretrained_requested = bool(pretrained)
        from rs_detection_amd.runner.checkpoint import load_pretrained
        m.pretrained_report = load_pretrained(m, name or "van", pretrained)   # warns loudly when nothing is available
        return m
    return make


van_b0 = BACKBONES.register_module(name="van_b0", module=_variant([32, 64, 160, 256], [8, 8, 4, 4], [3, 3, 5, 2], "van_b0"))
van_b1 = BACKBONES.register_module(name="van_b1", module=_variant([64, 128, 320, 512], [8, 8, 4, 4], [2, 2, 4, 2], "van_b1"))
van_b2 = BACKBONES.register_module(name="van_b2", module=_variant([64, 128, 320, 512], [8, 8, 4, 4], [3, 3, 12, 3], "van_b2"))
van_b3 = BACKBONES.register_module(name="van_b3", module=_variant([64, 128, 320, 512], [8, 8, 4, 4], [3, 5, 27, 3], "van_b3"))
