"""Box conversions on the S2ANet path (reference: /root/reference/python/jdet/models/boxes/box_ops.py).
The arithmetic lives in csrc/box_coder.hip; these are the reference-named entry points."""
import numpy as np
import torch

from rs_detection_amd.utils.consts import const_tensor
from rs_detection_amd.ops.box_coder import (bbox2delta_rotated, delta2bbox_rotated, rotated_box_to_poly)  # noqa: F401


def norm_angle(angle, angle_version='le135'):
    """box_ops.py:176-182 (torch remainder == Python-style mod)."""
    lo = float(-np.pi / 2) if angle_version == 'le90' else float(-np.pi / 4)
    return torch.remainder(angle - lo, float(np.pi)) + lo


# ---- horizontal / RetinaNet-style coders (box_ops.py:5-129, :691-716); pure torch, run on any device ----
def _safe_log(x, eps=1e-20):
    """jt.safe_log is un-vendored Jittor behaviour: log(max(x, eps)) adopted."""
    return torch.log(torch.clamp(x, min=eps))


def loc2bbox(src_bbox, loc, mean=(0., 0., 0., 0.), std=(1., 1., 1., 1.)):
    """box_ops.py:5-34: (x0,y0,x1,y1) anchors + (dx,dy,dw,dh) -> (x0,y0,x1,y1)."""
    if src_bbox.shape[0] == 0:
        return loc.new_zeros((0, 4))
    loc = loc * const_tensor(std, loc) + const_tensor(mean, loc)
    sw = src_bbox[:, 2:3] - src_bbox[:, 0:1]
    sh = src_bbox[:, 3:4] - src_bbox[:, 1:2]
    cx = loc[:, 0:1] * sw + src_bbox[:, 0:1] + 0.5 * sw
    cy = loc[:, 1:2] * sh + src_bbox[:, 1:2] + 0.5 * sh
    w, h = torch.exp(loc[:, 2:3]) * sw, torch.exp(loc[:, 3:4]) * sh
    return torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=1)


def loc2bbox_r(src_bbox, loc, mean=(0., 0., 0., 0., 0.), std=(1., 1., 1., 1., 1.)):
    """box_ops.py:36-64: (cx,cy,w,h,a) + (dx,dy,dw,dh,da) -> (cx,cy,w,h,a)."""
    if src_bbox.shape[0] == 0:
        return loc.new_zeros((0, 4))  # (sic) the reference returns 4 columns here
    loc = loc * const_tensor(std, loc) + const_tensor(mean, loc)
    sw, sh = src_bbox[:, 2:3], src_bbox[:, 3:4]
    cx = loc[:, 0:1] * sw + src_bbox[:, 0:1]
    cy = loc[:, 1:2] * sh + src_bbox[:, 1:2]
    w, h = torch.exp(loc[:, 2:3]) * sw, torch.exp(loc[:, 3:4]) * sh
    x1, y1, x2, y2 = cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1, loc[:, 4:5] + src_bbox[:, 4:5]], dim=1)


def bbox2loc_r(src_bbox, dst_bbox, mean=(0., 0., 0., 0., 0.), std=(1., 1., 1., 1., 1.)):
    """box_ops.py:66-86 (note the reference's `+ 1` on the anchor sides and the 1e-5 inside the log)."""
    w, h = src_bbox[:, 2:3], src_bbox[:, 3:4]
    dx = (dst_bbox[:, 0:1] - src_bbox[:, 0:1]) / (w + 1)
    dy = (dst_bbox[:, 1:2] - src_bbox[:, 1:2]) / (h + 1)
    dw = torch.log(dst_bbox[:, 2:3] / (w + 1) + 1e-5)
    dh = torch.log(dst_bbox[:, 3:4] / (h + 1) + 1e-5)
    da = dst_bbox[:, 4:5] - src_bbox[:, 4:5]
    loc = torch.cat([dx, dy, dw, dh, da], dim=1)
    return (loc - const_tensor(mean, loc)) / const_tensor(std, loc)


def bbox2loc(src_bbox, dst_bbox, mean=(0., 0., 0., 0.), std=(1., 1., 1., 1.)):
    """box_ops.py:88-115."""
    w = src_bbox[:, 2:3] - src_bbox[:, 0:1]
    h = src_bbox[:, 3:4] - src_bbox[:, 1:2]
    cx, cy = src_bbox[:, 0:1] + 0.5 * w, src_bbox[:, 1:2] + 0.5 * h
    bw = dst_bbox[:, 2:3] - dst_bbox[:, 0:1]
    bh = dst_bbox[:, 3:4] - dst_bbox[:, 1:2]
    bcx, bcy = dst_bbox[:, 0:1] + 0.5 * bw, dst_bbox[:, 1:2] + 0.5 * bh
    h, w = torch.clamp(h, min=1e-5), torch.clamp(w, min=1e-5)
    loc = torch.cat([(bcx - cx) / w, (bcy - cy) / h, _safe_log(bw / w), _safe_log(bh / h)], dim=1)
    return (loc - const_tensor(mean, loc)) / const_tensor(std, loc)


def bbox_iou(bbox_a, bbox_b):
    """box_ops.py:117-129: (N,4) x (K,4) x0y0x1y1 -> (N,K), no +1 convention."""
    assert bbox_a.shape[1] == 4 and bbox_b.shape[1] == 4
    if bbox_a.numel() == 0 or bbox_b.numel() == 0:
        return bbox_a.new_zeros((bbox_a.shape[0], bbox_b.shape[0]))
    tl = torch.maximum(bbox_a[:, None, :2], bbox_b[:, :2])
    br = torch.minimum(bbox_a[:, None, 2:], bbox_b[:, 2:])
    area_i = torch.prod(br - tl, dim=2) * (tl < br).all(dim=2)
    area_a = torch.prod(bbox_a[:, 2:] - bbox_a[:, :2], dim=1)
    area_b = torch.prod(bbox_b[:, 2:] - bbox_b[:, :2], dim=1)
    return area_i / (area_a[:, None] + area_b - area_i)


def rotated_box_to_poly_t(boxes):
    """Pure-torch form of rotated_box_to_poly (box_ops.py:633-654) for host-side target preparation."""
    cs, sn = torch.cos(boxes[:, 4]), torch.sin(boxes[:, 4])
    w, h = boxes[:, 2], boxes[:, 3]
    x_ctr, y_ctr = boxes[:, 0], boxes[:, 1]
    w_x, w_y, h_x, h_y = w / 2 * cs, w / 2 * sn, -h / 2 * sn, h / 2 * cs
    return torch.stack([x_ctr - w_x - h_x, y_ctr - w_y - h_y, x_ctr + w_x - h_x, y_ctr + w_y - h_y,
                        x_ctr + w_x + h_x, y_ctr + w_y + h_y, x_ctr - w_x + h_x, y_ctr - w_y + h_y], dim=1)


def rotated_box_to_bbox(rotated_boxes):
    """box_ops.py:691-697: axis-aligned hull (x0,y0,x1,y1) of rotated boxes."""
    polys = rotated_box_to_poly_t(rotated_boxes)
    xs, ys = polys[:, ::2], polys[:, 1::2]
    return torch.stack([xs.min(1)[0], ys.min(1)[0], xs.max(1)[0], ys.max(1)[0]], dim=1)


def boxes_xywh_to_x0y0x1y1(boxes):
    """box_ops.py:700-707."""
    assert boxes.shape[1] >= 4
    x, y, w, h = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    return torch.cat([torch.stack([x - 0.5 * w, y - 0.5 * h, x + 0.5 * w, y + 0.5 * h], dim=1), boxes[:, 4:]], dim=1)


def boxes_x0y0x1y1_to_xywh(boxes):
    """box_ops.py:709-716."""
    assert boxes.shape[1] >= 4
    x0, y0, x1, y1 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    return torch.cat([torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=1), boxes[:, 4:]], dim=1)
