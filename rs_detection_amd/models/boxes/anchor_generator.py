"""AnchorGeneratorRotatedS2ANet (/root/reference/python/jdet/models/boxes/anchor_generator.py:7-91).

Anchors are integers + 0.5 in fp32 (exact), generated once per (level, feature size)
on the target device and cached by the head."""
import torch

from rs_detection_amd.utils.registry import BOXES


@BOXES.register_module()
class AnchorGeneratorRotatedS2ANet:
    def __init__(self, base_size, scales, ratios, angles=[0, ], scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.tensor(scales, dtype=torch.float32)
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self.angles = torch.tensor(angles, dtype=torch.float32)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        w = h = self.base_size
        x_ctr, y_ctr = (0.5 * (w - 1), 0.5 * (h - 1)) if self.ctr is None else self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        assert self.scale_major, "AnchorGeneratorRotated only support scale-major anchors!"
        ones = torch.ones_like(self.angles)
        ws = (w * w_ratios[:, None, None] * self.scales[None, :, None] * ones[None, None, :]).reshape(-1)
        hs = (h * h_ratios[:, None, None] * self.scales[None, :, None] * ones[None, None, :]).reshape(-1)
        angles = self.angles.repeat(len(self.scales) * len(self.ratios))
        x = x_ctr + torch.zeros_like(ws)
        y = y_ctr + torch.zeros_like(ws)
        return torch.stack([x, y, ws, hs, angles], dim=-1)

    def grid_anchors(self, featmap_size, stride=16, device="cpu"):
        base = self.base_anchors.to(device)
        feat_h, feat_w = featmap_size
        sx = torch.arange(0, feat_w, device=device, dtype=torch.float32) * stride
        sy = torch.arange(0, feat_h, device=device, dtype=torch.float32) * stride
        xx = sx.repeat(feat_h)
        yy = sy.view(-1, 1).repeat(1, feat_w).view(-1)
        zeros = torch.zeros_like(xx)
        shifts = torch.stack([xx, yy, zeros, zeros, zeros], dim=-1)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 5)

    def valid_flags(self, featmap_size, valid_size, device="cpu"):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        vx = torch.zeros(feat_w, dtype=torch.bool, device=device)
        vy = torch.zeros(feat_h, dtype=torch.bool, device=device)
        vx[:valid_w] = True
        vy[:valid_h] = True
        valid = vx.repeat(feat_h) & vy.view(-1, 1).repeat(1, feat_w).view(-1)
        return valid[:, None].expand(valid.size(0), self.num_base_anchors).reshape(-1)
