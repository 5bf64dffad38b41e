"""anchor_target (/root/reference/python/jdet/models/boxes/anchor_target.py:18-195).

Two forms:
  * ``anchor_target`` / ``anchor_target_single``: the reference's per-image call
    structure and return tuple (host syncs at ``len(pos_inds)`` like the reference);
  * ``anchor_target_batched``: the MI355X-first form S2ANetHead uses -- all images in
    one grouped-IoU launch + one assignment call + a handful of elementwise torch ops,
    no host synchronisation (num_total_pos stays a device scalar).
Semantics kept: allowed_border=-1 => inside_flags = valid_flags (q10); PseudoSampler
when sampling=False; pos_weight<=0 => label weight 1; num_total_pos = sum_img max(#pos,1) (q11).
"""
import torch

from rs_detection_amd.utils.general import multi_apply, unmap
from rs_detection_amd.utils.registry import BOXES, build_from_cfg
from .sampler import PseudoSampler


def images_to_levels(target, num_level_anchors):
    target = torch.stack(target, 0) if isinstance(target, (list, tuple)) else target
    out, start = [], 0
    for n in num_level_anchors:
        out.append(target[:, start:start + n])
        start += n
    return out


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def _coder(cfg):
    c = cfg.get('bbox_coder', '')
    return build_from_cfg(dict(type='DeltaXYWHABBoxCoder') if c == '' else c, BOXES)


def anchor_target_single(flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels, img_meta, target_means,
                         target_stds, cfg=None, label_channels=1, sampling=True, unmap_outputs=True):
    bbox_coder = _coder(cfg)
    reg_decoded_bbox = cfg.get('reg_decoded_bbox', False)
    inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                       cfg.get('allowed_border', -1))
    if not bool(inside_flags.any()):
        return (None,) * 6
    anchors = flat_anchors[inside_flags, :]
    assert not sampling, "S2ANet/focal-loss path: sampling=False (PseudoSampler)"
    assigner = build_from_cfg(cfg.get('assigner', ''), BOXES)
    assign_result = assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore, gt_labels)
    sampling_result = PseudoSampler().sample(assign_result, anchors, gt_bboxes)
    n = anchors.shape[0]
    bbox_targets, bbox_weights = torch.zeros_like(anchors), torch.zeros_like(anchors)
    labels = torch.zeros(n, dtype=torch.int32, device=anchors.device)
    label_weights = torch.zeros(n, dtype=torch.float32, device=anchors.device)
    pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
    if len(pos_inds) > 0:
        tgt = sampling_result.pos_gt_bboxes if reg_decoded_bbox else \
            bbox_coder.encode(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes)
        bbox_targets[pos_inds, :] = tgt.to(bbox_targets.dtype)
        bbox_weights[pos_inds, :] = 1.0
        labels[pos_inds] = 1 if gt_labels is None else gt_labels[sampling_result.pos_assigned_gt_inds].to(labels.dtype)
        pw = cfg.get('pos_weight', -1)
        label_weights[pos_inds] = 1.0 if pw <= 0 else pw
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    if unmap_outputs:
        total = flat_anchors.size(0)
        labels = unmap(labels, total, inside_flags)
        label_weights = unmap(label_weights, total, inside_flags)
        bbox_targets = unmap(bbox_targets, total, inside_flags)
        bbox_weights = unmap(bbox_weights, total, inside_flags)
    return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds, cfg,
                  gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                  unmap_outputs=True):
    num_imgs = len(img_metas)
    assert len(anchor_list) == len(valid_flag_list) == num_imgs
    num_level_anchors = [a.size(0) for a in anchor_list[0]]
    anchor_list = [torch.cat(a) for a in anchor_list]
    valid_flag_list = [torch.cat(v) for v in valid_flag_list]
    if gt_bboxes_ignore_list is None:
        gt_bboxes_ignore_list = [None] * num_imgs
    if gt_labels_list is None:
        gt_labels_list = [None] * num_imgs
    (all_labels, all_label_weights, all_bbox_targets, all_bbox_weights, pos_inds_list, neg_inds_list) = multi_apply(
        anchor_target_single, anchor_list, valid_flag_list, gt_bboxes_list, gt_bboxes_ignore_list, gt_labels_list,
        img_metas, target_means=target_means, target_stds=target_stds, cfg=cfg, label_channels=label_channels,
        sampling=sampling, unmap_outputs=unmap_outputs)
    if any(l is None for l in all_labels):
        return None
    num_total_pos = sum(max(i.numel(), 1) for i in pos_inds_list)
    num_total_neg = sum(max(i.numel(), 1) for i in neg_inds_list)
    return (images_to_levels(all_labels, num_level_anchors), images_to_levels(all_label_weights, num_level_anchors),
            images_to_levels(all_bbox_targets, num_level_anchors),
            images_to_levels(all_bbox_weights, num_level_anchors), num_total_pos, num_total_neg)


def _fused_ok(assigner, coder, cfg):
    """The sparse two-launch path (csrc/anchor_target.hip) covers: a rotated IoU calculator, gt_max_assign_all, no
    ignore regions, the DeltaXYWHA coder.  Anything else takes the dense chain below (same results, more launches)."""
    import os
    from .coder import DeltaXYWHABBoxCoder
    if os.environ.get("RSDET_DENSE_ANCHOR_TARGET", "0") == "1":
        return False
    return (getattr(assigner.iou_calculator, "version", None) in (0, 1) and assigner.gt_max_assign_all
            and type(coder) is DeltaXYWHABBoxCoder)


def anchor_target_batched(anchors, gt_cat, gt_labels_cat, row_offsets, max_k, cfg, assigner=None, coder=None,
                          valid=None, ks=None, cache_anchors=False, heavy_from=None, prepared_gt=None):
    """anchors (A,5) shared or (B,A,5); gt_cat (sumK,5); gt_labels_cat (sumK,) int; row_offsets (B+1) i32.
    -> labels (B,A) i32, label_weights (B,A), bbox_targets (B,A,5), bbox_weights (B,A,5),
       num_total_pos (device scalar, float), num_total_neg (device scalar, float).

    ``ks`` (the gt counts as Python ints, which every caller has: they are tensor shapes) selects the fused sparse
    path: rotated IoU of the overlapping pairs only -> assignment -> encode -> weights / counts in TWO launches
    (ops/anchor_target.py), no (K, A) matrix.  ``cache_anchors``: the anchors are the same tensor every step (the FAM
    grid) -- their prepared form is kept; ``heavy_from``: first anchor index of the large pyramid levels (hint);
    ``prepared_gt``: ``ops.prepare_boxes(gt_cat)`` made once by a caller that assigns the same gts twice (FAM + ODM) --
    never cached: the gts of a step are never seen again."""
    assigner = assigner or build_from_cfg(cfg.get('assigner', ''), BOXES)
    coder = coder or _coder(cfg)
    B = row_offsets.numel() - 1
    if ks is not None and _fused_ok(assigner, coder, cfg) and gt_labels_cat is not None:
        from rs_detection_amd.ops import anchor_target as _at
        prep = _at.prepare_boxes(anchors, cache=cache_anchors, heavy_from=None if cache_anchors else heavy_from)
        out = _at.anchor_target_rotated(
            anchors, gt_cat, gt_labels_cat.to(torch.int32), row_offsets, list(ks), assigner.pos_iou_thr,
            assigner._neg(), assigner.min_pos_iou, assigner.match_low_quality, assigner.assigned_labels_filled,
            float(cfg.get('pos_weight', -1)), bool(cfg.get('reg_decoded_bbox', False)), coder.means, coder.stds,
            valid, assigner.iou_calculator.version, prepared=prep,
            prepared_gt=prepared_gt if prepared_gt is not None else (
                _at.prepare_boxes(gt_cat, heavy_from=gt_cat.shape[0]) if gt_cat.shape[0] else None))
        return (out["labels"], out["label_weights"], out["bbox_targets"], out["bbox_weights"], out["totals"][0],
                out["totals"][1])
    gt_inds, _, labels = assigner.assign_batch(anchors, gt_cat, row_offsets, max_k, gt_labels_cat, valid)
    A = gt_inds.shape[1]
    pos = gt_inds > 0
    neg = gt_inds == 0
    # gather the matched gt of every anchor (index 0 for non-positives; masked below)
    gidx = (gt_inds.long() - 1).clamp(min=0) + row_offsets[:-1].long()[:, None]
    gidx = gidx.clamp(max=max(gt_cat.shape[0] - 1, 0))
    matched = gt_cat[gidx.view(-1)]
    anc = anchors if anchors.dim() == 3 else anchors[None].expand(B, A, 5)
    if cfg.get('reg_decoded_bbox', False):
        tgt = matched.view(B, A, 5)
    else:
        tgt = coder.encode(anc.reshape(-1, 5), matched).view(B, A, 5)
    posf = pos[..., None].to(tgt.dtype)
    bbox_targets = torch.where(pos[..., None], tgt, torch.zeros_like(tgt))
    bbox_weights = posf.expand(B, A, 5)
    pw = cfg.get('pos_weight', -1)
    label_weights = neg.to(tgt.dtype) + pos.to(tgt.dtype) * (1.0 if pw <= 0 else pw)
    num_total_pos = pos.sum(1).clamp(min=1).sum().to(tgt.dtype)
    num_total_neg = neg.sum(1).clamp(min=1).sum().to(tgt.dtype)
    return labels, label_weights, bbox_targets, bbox_weights, num_total_pos, num_total_neg
