"""jdet.ops.nms_rotated on MI355X: nms_rotated / ml_nms_rotated / multiclass_nms_rotated.

Mirror of /root/reference/python/jdet/ops/nms_rotated.py:495-596.  Parity target is
the reference CPU path: suppression on ``ovr >= iou_threshold`` (:444), stable
descending score order, kept indices returned ASCENDING like ``jt.where(keep)`` (:525).
"""
import torch

from .. import _lib

__all__ = ["nms_rotated", "ml_nms_rotated", "multiclass_nms_rotated", "nms_rotated_keep_mask"]


def nms_rotated_keep_mask(dets, order, iou_threshold, box_length=None, ge=True, label_major=False):
    """Device twin of nms_rotated_cpu/_cuda (:495-512): dets (n,5|6), order (n) -> bool keep (n).
    ``label_major=True`` promises that ``order`` keeps equal labels contiguous and score-descending inside
    (``_label_major_order``): the label runs are then swept concurrently."""
    _lib.require_cuda_f32(dets)
    lib = _lib.load()
    dets = dets.contiguous()
    n = dets.shape[0]
    bl = dets.shape[1] if box_length is None else box_length
    assert bl in (5, 6) and dets.shape[1] == bl
    keep = torch.empty((n,), dtype=torch.uint8, device=dets.device)
    if n == 0:
        return keep.bool()
    order = order.to(torch.int32).contiguous()
    ws_bytes = lib.rsdet_nms_rotated_ws_size(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    rc = lib.rsdet_nms_rotated_f32(_lib.ptr(dets), n, bl, _lib.ptr(order), float(iou_threshold),
                                   int(bool(ge)) | (2 if (label_major and bl == 6) else 0),
                                   _lib.ptr(keep), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_nms_rotated_f32")
    return keep.bool()


def _order(scores):
    # Jittor argsort tie-break is unpinned; stable descending is adopted.
    return torch.argsort(scores, dim=0, descending=True, stable=True)


def _label_major_order(scores, labels):
    """Score-descending inside each label, labels in ascending blocks.  Boxes of different labels never interact
    (nms_rotated.py:285-286), so any order that keeps every label's boxes score-descending yields the keep set of
    the global score order; grouping them lets nms_mask skip every tile whose two 64-box blocks hold different
    labels (csrc/nms_rotated.hip)."""
    by_score = torch.argsort(scores, dim=0, descending=True, stable=True)
    return by_score[torch.argsort(labels[by_score], dim=0, stable=True)]


def ml_nms_rotated(dets, scores, labels, iou_threshold):
    """nms_rotated.py:514-525: class-aware NMS; returns ascending kept indices."""
    assert dets.numel() > 0 and dets.dim() == 2
    assert dets.dtype == scores.dtype
    dets6 = torch.cat([dets, labels.to(dets.dtype).unsqueeze(1)], dim=1)
    keep = nms_rotated_keep_mask(dets6, _label_major_order(scores, labels), iou_threshold, 6, label_major=True)
    return torch.where(keep)[0]


def nms_rotated(dets, scores, iou_threshold):
    """nms_rotated.py:527-538."""
    if dets.numel() == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dets.device)
    assert dets.dim() == 2
    assert dets.dtype == scores.dtype
    keep = nms_rotated_keep_mask(dets, _order(scores), iou_threshold, 5)
    return torch.where(keep)[0]


def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """nms_rotated.py:540-596: (n, #cls*5 | 5) boxes, (n, #cls+1) scores (col 0 = background)
    -> (dets (k,6), labels (k,) 0-based)."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 5)[:, 1:]
    else:
        bboxes = multi_bboxes[:, None].expand(multi_bboxes.shape[0], num_classes, 5)
    scores = multi_scores[:, 1:]
    valid_mask = scores > score_thr
    bboxes = bboxes[valid_mask]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid_mask]
    labels = valid_mask.nonzero()[:, 1]
    if bboxes.numel() == 0:
        return (torch.zeros((0, 6), device=multi_bboxes.device),
                torch.zeros((0,), dtype=torch.int32, device=multi_bboxes.device))
    nms_cfg_ = dict(nms_cfg)
    nms_cfg_.pop('type', 'nms')
    iou_thr = nms_cfg_.pop('iou_thr', 0.1)
    keep = ml_nms_rotated(bboxes, scores, labels, iou_thr)
    bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
    inds = torch.argsort(scores, descending=True, stable=True)
    if keep.size(0) > max_num:  # NB :590 compares against max_num even when it is -1
        inds = inds[:max_num]
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    return torch.cat([bboxes, scores[:, None]], 1), labels
