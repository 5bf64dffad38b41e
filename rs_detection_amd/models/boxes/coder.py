"""DeltaXYWHABBoxCoder (/root/reference/python/jdet/models/boxes/coder.py:76-141)."""
from rs_detection_amd.ops.box_coder import bbox2delta_rotated, delta2bbox_rotated
from rs_detection_amd.utils.consts import const_tensor
from rs_detection_amd.utils.registry import BOXES


@BOXES.register_module()
class DeltaXYWHABBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.), clip_border=True):
        self.means, self.stds, self.clip_border = target_means, target_stds, clip_border

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 5
        return bbox2delta_rotated(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox_rotated(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                                  self.clip_border)


import numpy as np  # noqa: E402
import torch  # noqa: E402

from rs_detection_amd.ops import orpn  # noqa: E402
from rs_detection_amd.ops.bbox_transforms import obb2hbb, obb2poly, rectpoly2obb, regular_theta, regular_obb  # noqa: E402


@BOXES.register_module()
class MidpointOffsetCoder:
    """Oriented RPN coder (/root/reference/python/jdet/models/boxes/coder.py:318-433): hbb anchor + obb gt ->
    (dx, dy, dw, dh, da, db) midpoint offsets; decode -> obb via the rectified parallelogram."""

    def __init__(self, target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.)):
        self.means, self.stds = target_means, target_stds

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        p, gt = bboxes.float(), gt_bboxes.float()
        px, py = (p[..., 0] + p[..., 2]) * 0.5, (p[..., 1] + p[..., 3]) * 0.5
        pw, ph = p[..., 2] - p[..., 0], p[..., 3] - p[..., 1]
        hbb, poly = obb2hbb(gt), obb2poly(gt)
        gx, gy = (hbb[..., 0] + hbb[..., 2]) * 0.5, (hbb[..., 1] + hbb[..., 3]) * 0.5
        gw, gh = hbb[..., 2] - hbb[..., 0], hbb[..., 3] - hbb[..., 1]
        x_coor, y_coor = poly[:, 0::2], poly[:, 1::2]
        y_min = y_coor.min(dim=1, keepdim=True)[0]
        x_max = x_coor.max(dim=1, keepdim=True)[0]
        ga = torch.where((y_coor - y_min).abs() > 0.1, -1000., x_coor).max(1)[0]
        gb = torch.where((x_coor - x_max).abs() > 0.1, -1000., y_coor).max(1)[0]
        deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph),
                              (ga - gx) / gw, (gb - gy) / gh], dim=-1)
        means = const_tensor(self.means, deltas).unsqueeze(0)
        stds = const_tensor(self.stds, deltas).unsqueeze(0)
        return (deltas - means) / stds

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        rep = pred_bboxes.size(1) // 6
        if rep == 1 and orpn.decode_applies(bboxes, pred_bboxes):      # one kernel (csrc/orpn.hip) for ~50 tensor operations
            return orpn.midpoint_offset_decode(bboxes, pred_bboxes, self.means, self.stds,
                                               float(np.abs(np.log(wh_ratio_clip))))
        d = pred_bboxes * const_tensor(self.stds, pred_bboxes).repeat(rep) + const_tensor(self.means, pred_bboxes).repeat(rep)
        dx, dy, dw, dh, da, db = (d[:, k::6] for k in range(6))
        max_ratio = float(np.abs(np.log(wh_ratio_clip)))
        dw, dh = dw.clamp(-max_ratio, max_ratio), dh.clamp(-max_ratio, max_ratio)
        px, py = ((bboxes[:, 0] + bboxes[:, 2]) * 0.5).unsqueeze(1), ((bboxes[:, 1] + bboxes[:, 3]) * 0.5).unsqueeze(1)
        pw, ph = (bboxes[:, 2] - bboxes[:, 0]).unsqueeze(1), (bboxes[:, 3] - bboxes[:, 1]).unsqueeze(1)
        gw, gh = pw * dw.exp(), ph * dh.exp()
        gx, gy = px + pw * dx, py + ph * dy
        x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
        da, db = da.clamp(-0.5, 0.5), db.clamp(-0.5, 0.5)
        ga, _ga, gb, _gb = gx + da * gw, gx - da * gw, gy + db * gh, gy - db * gh
        polys = torch.stack([ga, y1, x2, gb, _ga, y2, x1, _gb], dim=-1)
        center = torch.stack([gx, gy, gx, gy, gx, gy, gx, gy], dim=-1)
        cp = polys - center
        diag = torch.sqrt(cp[..., 0::2] ** 2 + cp[..., 1::2] ** 2)
        scale = diag.max(dim=-1, keepdim=True)[0] / diag
        cp = cp * scale.repeat_interleave(2, dim=-1)
        return rectpoly2obb(cp + center).flatten(-2)


@BOXES.register_module()
class OrientedDeltaXYWHTCoder:
    """Oriented R-CNN head coder (/root/reference/python/jdet/models/boxes/coder.py:435-513)."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.)):
        self.means, self.stds = target_means, target_stds

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 5
        px, py, pw, ph, pt = bboxes.float().unbind(dim=-1)
        gx, gy, gw, gh, gt = gt_bboxes.float().unbind(dim=-1)
        d1, d2 = regular_theta(gt - pt), regular_theta(gt - pt + np.pi / 2)
        first = d1.abs() < d2.abs()
        gw_r, gh_r = torch.where(first, gw, gh), torch.where(first, gh, gw)
        dtheta = torch.where(first, d1, d2)
        c, s = torch.cos(-pt), torch.sin(-pt)
        dx = (c * (gx - px) + s * (gy - py)) / pw
        dy = (-s * (gx - px) + c * (gy - py)) / ph
        deltas = torch.stack([dx, dy, torch.log(gw_r / pw), torch.log(gh_r / ph), dtheta], dim=-1)
        return (deltas - const_tensor(self.means, deltas).unsqueeze(0)) / const_tensor(self.stds, deltas).unsqueeze(0)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        rep = pred_bboxes.size(1) // 5
        d = pred_bboxes * const_tensor(self.stds, pred_bboxes).repeat(rep) + const_tensor(self.means, pred_bboxes).repeat(rep)
        dx, dy, dw, dh, dt = (d[:, k::5] for k in range(5))
        max_ratio = float(np.abs(np.log(wh_ratio_clip)))
        dw, dh = dw.clamp(-max_ratio, max_ratio), dh.clamp(-max_ratio, max_ratio)
        px, py, pw, ph, pt = (v.unsqueeze(1).expand_as(dx) for v in bboxes.unbind(dim=-1))
        c, s = torch.cos(-pt), torch.sin(-pt)
        gx = dx * pw * c - dy * ph * s + px
        gy = dx * pw * s + dy * ph * c + py
        new = torch.stack([gx, gy, pw * dw.exp(), ph * dh.exp(), regular_theta(dt + pt)], dim=-1)
        return regular_obb(new).view_as(pred_bboxes)
