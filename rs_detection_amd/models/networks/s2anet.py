"""S2ANet detector (/root/reference/python/jdet/models/networks/s2anet.py:7-38):
backbone -> neck -> bbox_head(features, targets); train mode returns the loss dict,
eval mode a list of (polys (n,8), scores, labels) per image."""
import torch.nn as nn

from rs_detection_amd.utils.registry import MODELS, BACKBONES, HEADS, NECKS, build_from_cfg


@MODELS.register_module()
class S2ANet(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None):
        super().__init__()
        self.backbone = build_from_cfg(backbone, BACKBONES)
        self.neck = build_from_cfg(neck, NECKS)
        self.bbox_head = build_from_cfg(bbox_head, HEADS)

    def train(self, mode=True):
        super().train(mode)
        self.backbone.train(mode)
        return self

    def forward(self, images, targets):
        features = self.backbone(images)
        if self.neck is not None:
            features = self.neck(features)
        return self.bbox_head(features, targets)
