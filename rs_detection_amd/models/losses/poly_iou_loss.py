"""Polygon IoU / GIoU losses over convex_sort (/root/reference/python/jdet/models/losses/poly_iou_loss.py:13-241).

Tensor code in torch, the hull ordering by the fused HIP kernel (ops/convex_sort.py).  The intersection polygon of two
quadrilaterals is found the way the reference does it: all 16 edge-edge intersection points + the vertices of each
polygon that lie inside the other (masks), ordered into a convex polygon by `convex_sort`, area by the shoelace
formula.  Gradients flow through the point coordinates; the ordering itself carries none.
"""
import torch
import torch.nn as nn

from rs_detection_amd.ops.bbox_transforms import bbox2type, get_bbox_areas
from rs_detection_amd.ops.convex_sort import convex_sort
from rs_detection_amd.utils.registry import LOSSES


def shoelace(pts):
    """:13-18 -- area of polygons (..., n, 2)."""
    roll_pts = torch.roll(pts, 1, dims=-2)
    xyxy = pts[..., 0] * roll_pts[..., 1] - roll_pts[..., 0] * pts[..., 1]
    return 0.5 * xyxy.sum(dim=-1).abs()


def convex_areas(pts, masks):
    """:21-39 -- area of the convex hull of the masked points of every set (nbs, npts, 2)."""
    nbs, npts, _ = pts.size()
    index = convex_sort(pts, masks).long()
    index = torch.where(index == -1, torch.full_like(index, npts), index)   # unused slots -> the appended origin
    index = index[..., None].repeat(1, 1, 2)
    ext_pts = torch.cat([pts, pts.new_zeros((nbs, 1, 2))], dim=1)
    polys = torch.gather(ext_pts, 1, index)
    xyxy = polys[:, 0:-1, 0] * polys[:, 1:, 1] - polys[:, 0:-1, 1] * polys[:, 1:, 0]
    return 0.5 * xyxy.sum(dim=-1).abs()


def poly_intersection(pts1, pts2, areas1=None, areas2=None, eps=1e-6):
    """:42-92 -- candidate vertices of the intersection of two polygons and the mask of the real ones."""
    lines1 = torch.cat([pts1, torch.roll(pts1, -1, dims=1)], dim=2)
    lines2 = torch.cat([pts2, torch.roll(pts2, -1, dims=1)], dim=2)
    lines1, lines2 = lines1.unsqueeze(2), lines2.unsqueeze(1)
    x1, y1, x2, y2 = lines1.unbind(dim=-1)  # (N, 4, 1)
    x3, y3, x4, y4 = lines2.unbind(dim=-1)  # (N, 1, 4)

    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    with torch.no_grad():
        den_u = (x2 - x1) * (y1 - y3) - (y2 - y1) * (x1 - x3)
        t, u = den_t / num, den_u / num
        mask_inter = (t > 0) & (t < 1) & (u > 0) & (u < 1)

    t = den_t / (num + eps)
    pts_inter = torch.stack([x1 + t * (x2 - x1), y1 + t * (y2 - y1)], dim=-1)
    B = pts1.size(0)
    pts_inter = pts_inter.view(B, -1, 2)
    mask_inter = mask_inter.view(B, -1)

    # a vertex lies inside the other polygon iff the triangles it spans with that polygon's edges add up to its area
    with torch.no_grad():
        areas1 = shoelace(pts1) if areas1 is None else areas1
        areas2 = shoelace(pts2) if areas2 is None else areas2
        triangle_areas1 = 0.5 * ((x3 - x1) * (y4 - y1) - (y3 - y1) * (x4 - x1)).abs()
        sum_areas1 = triangle_areas1.sum(dim=-1)
        mask_inside1 = (sum_areas1 - areas2[..., None]).abs() < 1e-3 * areas2[..., None]
        triangle_areas2 = 0.5 * ((x1 - x3) * (y2 - y3) - (x2 - x3) * (y1 - y3)).abs()
        sum_areas2 = triangle_areas2.sum(dim=-2)
        mask_inside2 = (sum_areas2 - areas1[..., None]).abs() < 1e-3 * areas1[..., None]

    all_pts = torch.cat([pts_inter, pts1, pts2], dim=1)
    masks = torch.cat([mask_inter, mask_inside1, mask_inside2], dim=1)
    return all_pts, masks


def poly_enclose(pts1, pts2):
    """:95-100 -- the points whose hull encloses both polygons."""
    all_pts = torch.cat([pts1, pts2], dim=1)
    masks = all_pts.new_ones((all_pts.size(0), all_pts.size(1)))
    return all_pts, masks


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        avg_factor = loss.numel()
    if reduction == "sum":
        return loss.sum()
    if reduction == "mean":
        return loss.sum() / avg_factor
    return loss


def poly_overlaps(pred, target, eps=1e-6):
    """IoU, union, and the two point sets of aligned boxes (any of hbb / obb / poly layouts) -- the shared front
    half of poly_iou_loss / poly_giou_loss (:103-112, :131-142)."""
    areas1, areas2 = get_bbox_areas(pred), get_bbox_areas(target)
    pred, target = bbox2type(pred, 'poly'), bbox2type(target, 'poly')
    pred_pts = pred.view(pred.size(0), -1, 2)
    target_pts = target.view(target.size(0), -1, 2)
    inter_pts, inter_masks = poly_intersection(pred_pts, target_pts, areas1, areas2, eps)
    overlap = convex_areas(inter_pts, inter_masks)
    union = areas1 + areas2 - overlap + eps
    return overlap / union, union, pred_pts, target_pts


def poly_iou_loss(pred, target, linear=False, eps=1e-6, weight=None, reduction='mean', avg_factor=None):
    """:103-128."""
    ious, _, _, _ = poly_overlaps(pred, target, eps)
    ious = ious.clamp(min=eps)
    loss = 1 - ious if linear else -ious.log()
    return _reduce(loss, weight, reduction, avg_factor)


def poly_giou_loss(pred, target, eps=1e-6, weight=None, reduction='mean', avg_factor=None):
    """:131-157."""
    ious, union, pred_pts, target_pts = poly_overlaps(pred, target, eps)
    ious = ious.clamp(min=eps)
    enclose_pts, enclose_masks = poly_enclose(pred_pts, target_pts)
    enclose_areas = convex_areas(enclose_pts, enclose_masks)
    gious = ious - (enclose_areas - union) / enclose_areas
    return _reduce(1 - gious, weight, reduction, avg_factor)


class _PolyLossBase(nn.Module):
    def _weight(self, pred, weight):
        if weight is not None and weight.dim() > 1:  # (n, k) weights -> (n,) like the loss (:190-195)
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return weight


@LOSSES.register_module()
class PolyIoULoss(_PolyLossBase):
    """:160-202."""

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.linear, self.eps, self.reduction, self.loss_weight = linear, eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * poly_iou_loss(pred, target, weight=self._weight(pred, weight), linear=self.linear,
                                                eps=self.eps, reduction=reduction, avg_factor=avg_factor, **kwargs)


@LOSSES.register_module()
class PolyGIoULoss(_PolyLossBase):
    """:205-241.  (The reference passes `weight` positionally into `poly_giou_loss`'s `eps` slot, :234-237, which
    raises as soon as the loss is called; here it goes by keyword.)"""

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and not bool((weight > 0).any()) and reduction != 'none':
            return (pred * (weight if weight.dim() == pred.dim() else weight[..., None])).sum()  # 0 (:224-226)
        return self.loss_weight * poly_giou_loss(pred, target, weight=self._weight(pred, weight), eps=self.eps,
                                                 reduction=reduction, avg_factor=avg_factor, **kwargs)
