"""RCNN / OrientedRCNN detectors (/root/reference/python/jdet/models/networks/rcnn.py:8-56,
oriented_rcnn.py:5-9): backbone -> neck -> rpn(features, targets) -> bbox_head(features, proposals, targets)."""
import torch.nn as nn

from rs_detection_amd.utils.registry import MODELS, BACKBONES, HEADS, NECKS, build_from_cfg


@MODELS.register_module()
class RCNN(nn.Module):
    def __init__(self, backbone, neck=None, rpn=None, bbox_head=None):
        super().__init__()
        self.backbone = build_from_cfg(backbone, BACKBONES)
        self.neck = build_from_cfg(neck, NECKS)
        self.rpn = build_from_cfg(rpn, HEADS)
        self.bbox_head = build_from_cfg(bbox_head, HEADS)

    def forward(self, images, targets):
        features = self.backbone(images)
        if self.neck is not None:
            features = self.neck(features)
        proposals_list, rpn_losses = self.rpn(features, targets)
        output = self.bbox_head(features, proposals_list, targets)
        if self.training:
            output.update(rpn_losses)
        return output


@MODELS.register_module()
class OrientedRCNN(RCNN):
    pass
