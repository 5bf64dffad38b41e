"""FPN neck (/root/reference/python/jdet/models/necks/fpn.py:9-202).

``start_level``, ``add_extra_convs='on_input'|'on_lateral'|'on_output'``, nearest
upsample top-down path, extra stride-2 levels WITHOUT ReLU unless
``relu_before_extra_convs``; xavier-uniform init."""
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.utils.registry import NECKS
from rs_detection_amd.models.utils.modules import ConvModule
from rs_detection_amd.models.utils.weight_init import xavier_init


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=None,
                 upsample_div_factor=1):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        self.upsample_div_factor = upsample_div_factor
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        self.add_extra_convs = add_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg, act_cfg=act_cfg))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg))
        extra = num_outs - self.backbone_end_level + start_level
        if add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and add_extra_convs == 'on_input') else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        n = len(lat)
        for i in range(n - 1, 0, -1):
            if 'scale_factor' in self.upsample_cfg:
                up = F.interpolate(lat[i], **self.upsample_cfg)
            else:
                up = F.interpolate(lat[i], size=lat[i - 1].shape[2:], **self.upsample_cfg)
            lat[i - 1] = lat[i - 1] + up
            if self.upsample_div_factor != 1:
                lat[i - 1] = lat[i - 1] / self.upsample_div_factor
        outs = [self.fpn_convs[i](lat[i]) for i in range(n)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - n):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                src = {'on_input': inputs[self.backbone_end_level - 1], 'on_lateral': lat[-1],
                       'on_output': outs[-1]}.get(self.add_extra_convs if isinstance(self.add_extra_convs, str) else 'on_input')
                outs.append(self.fpn_convs[n](src))
                for i in range(n + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)
