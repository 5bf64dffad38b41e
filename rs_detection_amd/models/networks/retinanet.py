"""RetinaNet detector (/root/reference/python/jdet/models/networks/retinanet.py:9-60): backbone -> neck ->
``rpn_net`` (a RetinaHead).  When the targets carry ``rboxes`` (le135, [-pi/4, 3pi/4)) they are first folded to
[-pi/2, 0) and w >= h, and their axis-aligned hulls are added as ``rboxes_h`` (:30-51); train mode returns the
loss dict, eval mode a list of (polys, scores, labels) per image."""
import numpy as np
import torch
import torch.nn as nn

from rs_detection_amd.utils.registry import MODELS, BACKBONES, HEADS, NECKS, build_from_cfg
from rs_detection_amd.models.boxes.box_ops import rotated_box_to_bbox


@MODELS.register_module()
class RetinaNet(nn.Module):
    def __init__(self, backbone, neck=None, rpn_net=None):
        super().__init__()
        self.backbone = build_from_cfg(backbone, BACKBONES)
        self.neck = build_from_cfg(neck, NECKS)
        self.rpn_net = build_from_cfg(rpn_net, HEADS)

    def train(self, mode=True):
        super().train(mode)
        self.backbone.train(mode)
        return self

    @staticmethod
    def fold_angles(rboxes):
        """:33-45, vectorised: a >= 0 -> a - pi; then a < -pi/2 -> a + pi/2 with w, h swapped."""
        x, y, w, h, a = rboxes.unbind(1)
        a = torch.where(a >= 0, a - np.pi, a)
        swap = a < -np.pi / 2
        a = torch.where(swap, a + np.pi / 2, a)
        return torch.stack([x, y, torch.where(swap, h, w), torch.where(swap, w, h), a], dim=1)

    def forward(self, images, targets):
        if "rboxes" in targets[0] and getattr(self.rpn_net, "mode", 'R') != 'H':
            targets = [dict(t) for t in targets]  # the reference rewrites the caller's dicts in place
            for t in targets:
                temp = self.rpn_net.cvt2_w_greater_than_h(self.fold_angles(t["rboxes"][:, :5]), False)
                t["rboxes"] = temp
                temp_ = temp.clone()
                temp_[:, 4] += np.pi / 2
                t["rboxes_h"] = rotated_box_to_bbox(temp_)
        features = self.backbone(images)
        if self.neck is not None:
            features = self.neck(features)
        results, losses = self.rpn_net(features, targets)
        return losses if self.training else results
