"""Device twins of the reference's elementwise box code + the batched assigner.

bbox2delta_rotated / delta2bbox_rotated : models/boxes/box_ops.py:184-289
s2a_refine_and_offset                   : roi_heads/s2anet_head.py:631-654 + :676-713 (fused)
rotated_box_to_poly                     : models/boxes/box_ops.py:633-654
assign_wrt_overlaps                     : models/boxes/assigner.py:111-170 (whole batch, one call)
(paths relative to /root/reference/python/jdet/; kernels: csrc/box_coder.hip, csrc/assign.hip)
"""
import ctypes
import math

import torch

from .. import _lib


def bbox2delta_rotated(proposals, gt, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.)):
    assert proposals.size() == gt.size()
    _lib.require_cuda_f32(proposals, gt)
    lib = _lib.load()
    p, g = proposals.contiguous().view(-1, 5), gt.contiguous().view(-1, 5)
    out = torch.empty_like(p)
    rc = lib.rsdet_bbox2delta_rotated_f32(_lib.ptr(p), _lib.ptr(g), p.shape[0], _lib.host5(means, 0.),
                                          _lib.host5(stds, 1.), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rsdet_bbox2delta_rotated_f32")
    return out.view(proposals.shape)


def delta2bbox_rotated(rois, deltas, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.), max_shape=None,
                       wh_ratio_clip=16 / 1000, clip_border=True):
    """max_shape / clip_border are accepted and ignored, as in the reference."""
    _lib.require_cuda_f32(rois, deltas)
    assert deltas.size(1) == 5, "single-class (N,5) deltas"
    lib = _lib.load()
    r, d = rois.contiguous(), deltas.contiguous()
    out = torch.empty_like(d)
    max_ratio = abs(math.log(wh_ratio_clip))
    rc = lib.rsdet_delta2bbox_rotated_f32(_lib.ptr(r), _lib.ptr(d), r.shape[0], _lib.host5(means, 0.),
                                          _lib.host5(stds, 1.), max_ratio, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rsdet_delta2bbox_rotated_f32")
    return out


def s2a_refine_and_offset(bbox_pred, anchors, stride, kernel_size=3, means=(0.,) * 5, stds=(1.,) * 5,
                          wh_ratio_clip=1e-6, want_offset=True):
    """bbox_pred (B,5,H,W), anchors (H*W,5) -> refined (B,H,W,5), offset (B,2*ks*ks,H,W)."""
    _lib.require_cuda_f32(bbox_pred, anchors)
    lib = _lib.load()
    bp, an = bbox_pred.contiguous(), anchors.contiguous()
    B, _, H, W = bp.shape
    refined = torch.empty((B, H, W, 5), dtype=bp.dtype, device=bp.device)
    offset = torch.empty((B, 2 * kernel_size * kernel_size, H, W), dtype=bp.dtype, device=bp.device) if want_offset else None
    rc = lib.rsdet_s2a_refine_and_offset_f32(_lib.ptr(bp), _lib.ptr(an), B, H, W, float(stride), kernel_size,
                                             _lib.host5(means, 0.), _lib.host5(stds, 1.),
                                             abs(math.log(wh_ratio_clip)), _lib.ptr(refined), _lib.ptr(offset),
                                             _lib.stream_ptr())
    _lib.check(rc, "rsdet_s2a_refine_and_offset_f32")
    return refined, offset


def s2a_refine_and_offset_levels(bbox_preds, anchors, strides, kernel_size=3, means=(0.,) * 5, stds=(1.,) * 5,
                                 wh_ratio_clip=1e-6, want_offset=True):
    """s2a_refine_and_offset for all pyramid levels in ONE launch (rsdet_s2a_refine_and_offset_multi): lists of
    bbox_pred (B,5,H,W) -- fp32 or, from an autocast step, bf16 (widened inside the kernel: no cast pass) -- and anchors
    (H*W,5) fp32 -> lists of refined (B,H,W,5) and offset (B,2*ks*ks,H,W), fp32, the values of the per-level calls."""
    lib = _lib.load()
    n = len(bbox_preds)
    dt = bbox_preds[0].dtype
    if not (0 < n <= 8 and dt in (torch.float32, torch.bfloat16) and all(p.dtype == dt and p.is_cuda for p in bbox_preds)):
        raise ValueError("s2a_refine_and_offset_levels: 1..8 CUDA levels of one dtype (fp32 / bf16)")
    preds = [p.contiguous() for p in bbox_preds]
    ancs = [a.contiguous() for a in anchors]
    _lib.require_cuda_f32(*ancs)
    B = preds[0].shape[0]
    lv = _lib.S2aLevels()
    lv.n_levels, lv.B, lv.ks, lv.pred_bf16 = n, B, kernel_size, int(dt == torch.bfloat16)
    refined, offsets = [], []
    for l, (p, a, s) in enumerate(zip(preds, ancs, strides)):
        _, _, H, W = p.shape
        r = torch.empty((B, H, W, 5), dtype=torch.float32, device=p.device)
        o = torch.empty((B, 2 * kernel_size * kernel_size, H, W), dtype=torch.float32, device=p.device) if want_offset else None
        lv.H[l], lv.W[l], lv.stride[l] = H, W, float(s)
        lv.pred[l], lv.anchors[l], lv.refined[l], lv.offset[l] = _lib.ptr(p), _lib.ptr(a), _lib.ptr(r), _lib.ptr(o)
        refined.append(r)
        offsets.append(o)
    m, sd = _lib.host5(means, 0.), _lib.host5(stds, 1.)
    lv.means, lv.stds = ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p)
    lv.max_ratio = abs(math.log(wh_ratio_clip))
    _lib.check(lib.rsdet_s2a_refine_and_offset_multi(lv, _lib.stream_ptr()), "rsdet_s2a_refine_and_offset_multi")
    return refined, offsets


def rotated_box_to_poly(rrects):
    n = rrects.shape[0]
    if n == 0:
        return torch.zeros((0, 8), device=rrects.device)
    _lib.require_cuda_f32(rrects)
    lib = _lib.load()
    r = rrects.contiguous()
    out = torch.empty((n, 8), dtype=r.dtype, device=r.device)
    _lib.check(lib.rsdet_rotated_box_to_poly_f32(_lib.ptr(r), n, _lib.ptr(out), _lib.stream_ptr()),
               "rsdet_rotated_box_to_poly_f32")
    return out


def assign_wrt_overlaps(overlaps, row_offsets, max_rows, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0,
                        match_low_quality=True, gt_max_assign_all=True, gt_labels=None, labels_filled=0):
    """overlaps (n1, A) rows grouped by row_offsets (G+1 int32, device) ->
    gt_inds (G,A) int32, max_overlaps (G,A), labels (G,A) int32 | None."""
    _lib.require_cuda_f32(overlaps)
    lib = _lib.load()
    ov = overlaps.contiguous()
    n1, A = ov.shape
    G = row_offsets.numel() - 1
    dev = ov.device
    gt_inds = torch.empty((G, A), dtype=torch.int32, device=dev)
    max_ov = torch.empty((G, A), dtype=torch.float32, device=dev)
    labels = torch.empty((G, A), dtype=torch.int32, device=dev) if gt_labels is not None else None
    gl = gt_labels.to(torch.int32).contiguous() if gt_labels is not None else None
    if isinstance(neg_iou_thr, (tuple, list)):
        neg_lo, neg_hi = neg_iou_thr
    else:
        neg_lo, neg_hi = 0.0, neg_iou_thr
    ws_bytes = lib.rsdet_assign_ws_size(n1)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_assign_wrt_overlaps_f32(_lib.ptr(ov), n1, A, _lib.ptr(row_offsets), G, int(max_rows),
                                           float(pos_iou_thr), float(neg_lo), float(neg_hi), float(min_pos_iou),
                                           int(bool(match_low_quality)), int(bool(gt_max_assign_all)),
                                           _lib.ptr(gl), int(labels_filled), _lib.ptr(gt_inds), _lib.ptr(max_ov),
                                           _lib.ptr(labels), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_assign_wrt_overlaps_f32")
    return gt_inds, max_ov, labels
