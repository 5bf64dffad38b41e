"""MaxIoUAssigner (/root/reference/python/jdet/models/boxes/assigner.py:18-170).

``assign`` keeps the reference's per-image signature and raise behaviour; the
thresholding + per-gt low-quality loop (:125-168) runs in ONE device call
(csrc/assign.hip) with no host sync.  ``assign_batch`` does a whole batch:
grouped rotated IoU + assignment, two launches + two passes for all images."""
import torch

from rs_detection_amd import ops
from rs_detection_amd.utils.consts import const_tensor
from rs_detection_amd.utils.registry import BOXES, build_from_cfg


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    def add_gt_(self, gt_labels):
        n = len(gt_labels)
        self_inds = torch.arange(1, n + 1, dtype=self.gt_inds.dtype, device=self.gt_inds.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(self.num_gts), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels.to(self.labels.dtype), self.labels])


@BOXES.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, assigned_labels_filled=0,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.match_low_quality = ignore_wrt_candidates, match_low_quality
        self.assigned_labels_filled = assigned_labels_filled
        self.iou_calculator = build_from_cfg(iou_calculator, BOXES)

    def _neg(self):
        if isinstance(self.neg_iou_thr, (tuple, list)):
            assert len(self.neg_iou_thr) == 2
            return tuple(self.neg_iou_thr)
        return float(self.neg_iou_thr)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        no_ignore = not (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0)
        if no_ignore and type(self.iou_calculator).__name__ == "BboxOverlaps2D":
            # horizontal boxes (the Oriented RPN's 611 072 anchors per tile): both passes recompute the IoU instead of
            # writing and re-reading a (K, A) matrix (csrc/orpn.hip: rsdet_hbb_assign_f32; same gt_inds bit for bit)
            from rs_detection_amd.ops import orpn
            b4, g4 = bboxes[..., :4], gt_bboxes[..., :4]
            if orpn.hbb_assign_applies(b4, g4):
                gi, mo = orpn.hbb_assign(b4, g4, self.pos_iou_thr, self._neg(), self.min_pos_iou, self.match_low_quality,
                                         self.gt_max_assign_all)
                labels = None
                if gt_labels is not None:
                    lab = gt_labels.to(torch.int32)
                    labels = torch.where(gi > 0, lab[(gi - 1).clamp(min=0).long()],
                                         torch.full_like(gi, self.assigned_labels_filled))
                return AssignResult(gt_bboxes.shape[0], gi, mo, labels)
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            if self.ignore_wrt_candidates:
                ign = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                ign = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
            overlaps[:, ign > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        if overlaps.numel() == 0:
            raise ValueError('No gt or proposals')
        K = overlaps.size(0)
        ro = const_tensor([0, K], dtype=torch.int32, device=overlaps.device)
        gi, mo, lb = ops.assign_wrt_overlaps(overlaps, ro, K, self.pos_iou_thr, self._neg(), self.min_pos_iou,
                                             self.match_low_quality, self.gt_max_assign_all, gt_labels,
                                             self.assigned_labels_filled)
        return AssignResult(K, gi[0], mo[0], None if lb is None else lb[0])

    # ---- batched (MI355X-first) form -------------------------------------------------
    def assign_batch(self, bboxes, gt_bboxes_cat, row_offsets, max_k, gt_labels_cat=None, valid=None):
        """bboxes (A,5) shared or (B,A,5) per image; gt_bboxes_cat (sumK,5);
        row_offsets (B+1) int32 device; -> gt_inds (B,A) i32, max_overlaps (B,A), labels (B,A)|None.
        ``valid`` (B,A) bool: anchors outside it are taken out of the assignment like the
        reference's ``anchors[inside_flags]`` subsetting (anchor_target.py:124-130)."""
        version = getattr(self.iou_calculator, "version", None)
        assert version is not None, "assign_batch needs a rotated IoU calculator"
        ov = ops.box_iou_rotated_grouped(gt_bboxes_cat, row_offsets, max_k, bboxes, version)
        if valid is not None:
            B = row_offsets.numel() - 1
            rows = torch.arange(ov.shape[0], device=ov.device)
            grp = torch.bucketize(rows, row_offsets[1:].long(), right=True).clamp(max=B - 1)
            ov = torch.where(valid[grp], ov, -1.0)
        return ops.assign_wrt_overlaps(ov, row_offsets, max_k, self.pos_iou_thr, self._neg(), self.min_pos_iou,
                                       self.match_low_quality, self.gt_max_assign_all, gt_labels_cat,
                                       self.assigned_labels_filled)
