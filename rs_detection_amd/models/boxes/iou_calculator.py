"""IoU calculators registered in BOXES (/root/reference/python/jdet/models/boxes/iou_calculator.py:79-162)."""
import torch

from rs_detection_amd.ops import box_iou_rotated, box_iou_rotated_v1
from rs_detection_amd.utils.registry import BOXES


def bbox_overlaps_rotated(rboxes1, rboxes2, version=0):
    fn = box_iou_rotated if version == 0 else box_iou_rotated_v1
    return fn(rboxes1.float(), rboxes2.float())


class _Rotated:
    version = 0

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 5, 6]
        assert bboxes2.size(-1) in [0, 5, 6]
        assert mode == "iou" and is_aligned is False
        # a trailing score column is ignored by the kernel (row stride 6)
        return bbox_overlaps_rotated(bboxes1, bboxes2, self.version)

    def __repr__(self):
        return self.__class__.__name__ + '()'


@BOXES.register_module()
class BboxOverlaps2D_rotated(_Rotated):
    version = 0


@BOXES.register_module()
class BboxOverlaps2D_rotated_v1(_Rotated):
    version = 1


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """Horizontal-box IoU / IoF (pure tensor math, as in the reference :164-257)."""
    assert mode in ['iou', 'iof']
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows,) if is_aligned else (rows, cols))
    a1 = (bboxes1[:, 2] - bboxes1[:, 0]) * (bboxes1[:, 3] - bboxes1[:, 1])
    a2 = (bboxes2[:, 2] - bboxes2[:, 0]) * (bboxes2[:, 3] - bboxes2[:, 1])
    if (not is_aligned and bboxes1.is_cuda and bboxes1.dtype == torch.float32 and bboxes2.dtype == torch.float32
            and bboxes1.dim() == 2 and bboxes2.dim() == 2 and not (bboxes1.requires_grad or bboxes2.requires_grad)):
        # one pass instead of ten over (rows, cols[, 2]) intermediates (csrc/assign.hip: bbox_overlaps_kernel): the same
        # operations in the same order, bit-identical; the Oriented RPN assigns 400 gts to 400 000 anchors with this
        from rs_detection_amd import _lib
        b1 = bboxes1 if bboxes1.stride(1) == 1 else bboxes1.contiguous()
        b2 = bboxes2 if bboxes2.stride(1) == 1 else bboxes2.contiguous()
        out = torch.empty((rows, cols), dtype=torch.float32, device=bboxes1.device)
        rc = _lib.load().rsdet_bbox_overlaps_f32(_lib.ptr(b1), rows, b1.stride(0), _lib.ptr(b2), cols, b2.stride(0),
                                                 int(mode == 'iof'), float(eps), _lib.ptr(out), _lib.stream_ptr())
        if rc == _lib.RSDET_OK:
            return out
    if is_aligned:
        lt = torch.max(bboxes1[:, :2], bboxes2[:, :2])
        rb = torch.min(bboxes1[:, 2:4], bboxes2[:, 2:4])
        wh = (rb - lt).clamp(min=0)
        ov = wh[:, 0] * wh[:, 1]
        union = a1 + a2 - ov if mode == 'iou' else a1
    else:
        lt = torch.max(bboxes1[:, None, :2], bboxes2[None, :, :2])
        rb = torch.min(bboxes1[:, None, 2:4], bboxes2[None, :, 2:4])
        wh = (rb - lt).clamp(min=0)
        ov = wh[..., 0] * wh[..., 1]
        union = a1[:, None] + a2[None, :] - ov if mode == 'iou' else a1[:, None].expand_as(ov)
    return ov / union.clamp_min(eps)


@BOXES.register_module()
class BboxOverlaps2D:
    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5]
        assert bboxes2.size(-1) in [0, 4, 5]
        return bbox_overlaps(bboxes1[..., :4], bboxes2[..., :4], mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'
