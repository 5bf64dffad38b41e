"""jdet.ops.box_iou_rotated / box_iou_rotated_v1 on MI355X.

Mirror of /root/reference/python/jdet/ops/box_iou_rotated.py:502-509 and
box_iou_rotated_v1.py:507-524 (same names, argument meaning, error behaviour);
the arithmetic runs in rs_detection_amd/csrc/box_iou_rotated.hip.
"""
import torch

from .. import _lib

__all__ = ["box_iou_rotated", "box_iou_rotated_v1", "box_iou_rotated_grouped"]


def _iou(boxes1, boxes2, version):
    assert boxes1.dtype == boxes2.dtype  # box_iou_rotated.py:503
    _lib.require_cuda_f32(boxes1, boxes2)
    lib = _lib.load()
    b1, b2 = boxes1.contiguous(), boxes2.contiguous()
    n1, n2 = b1.shape[0], b2.shape[0]
    ious = torch.empty((n1, n2), dtype=torch.float32, device=b1.device)
    if n1 and n2:
        assert b1.dim() == 2 and b2.dim() == 2 and b1.shape[1] >= 5 and b2.shape[1] >= 5
        ws_bytes = lib.rsdet_box_iou_rotated_ws_size(n1, n2, n2)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b1.device)
        rc = lib.rsdet_box_iou_rotated_f32(_lib.ptr(b1), n1, b1.shape[1], _lib.ptr(b2), n2, b2.shape[1], version,
                                           _lib.ptr(ious), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, "rsdet_box_iou_rotated_f32")
    return ious


def box_iou_rotated(boxes1, boxes2):
    """(n1,5),(n2,5) [cx,cy,w,h,theta rad] -> (n1,n2) IoU."""
    return _iou(boxes1, boxes2, 0)


def box_iou_rotated_v1(boxes1, boxes2):
    """y-down angle convention (box_iou_rotated_v1.py:69-72) + the wrapper's
    "too small" filter (:515-522).  The reference writes ``min(1)[0]``, which in
    Jittor indexes the FIRST BOX's min side only; reproduced literally:
    when box 0 of either set has a side < 1e-3 ... the masks are scalars, so
    ``nonzero`` yields index 0 and row/column 0 is zeroed."""
    ious = _iou(boxes1, boxes2, 1)
    if boxes1.shape[0] and boxes2.shape[0]:
        small1 = boxes1[0, 2:4].min() < 0.001
        small2 = boxes2[0, 2:4].min() < 0.001
        # device-side, no host sync: scale row 0 / column 0 by the flags
        ious[0, :] = torch.where(small1, torch.zeros_like(ious[0, :]), ious[0, :])
        ious[:, 0] = torch.where(small2, torch.zeros_like(ious[:, 0]), ious[:, 0])
    return ious


def box_iou_rotated_grouped(boxes1, row_offsets, max_rows, boxes2, version=0, out=None):
    """Batched form: rows [row_offsets[g], row_offsets[g+1]) of ``boxes1`` against
    ``boxes2[g]`` (boxes2 (G,A,5)) or a shared ``boxes2`` (A,5).  One launch for the
    whole batch (replaces the per-image loop anchor_target.py:60-72)."""
    _lib.require_cuda_f32(boxes1, boxes2)
    lib = _lib.load()
    b1, b2 = boxes1.contiguous(), boxes2.contiguous()
    G = row_offsets.numel() - 1
    n1 = b1.shape[0]
    if b2.dim() == 3:
        assert b2.shape[0] == G
        A, gs = b2.shape[1], b2.shape[1] * b2.shape[2]
    else:
        A, gs = b2.shape[0], 0
    ious = out if out is not None else torch.empty((n1, A), dtype=torch.float32, device=b1.device)
    assert row_offsets.dtype == torch.int32 and row_offsets.is_cuda
    if n1 == 0 or A == 0:
        return ious
    ws_bytes = lib.rsdet_box_iou_rotated_ws_size(n1, G * A if gs else A, A)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b1.device)
    rc = lib.rsdet_box_iou_rotated_grouped_f32(_lib.ptr(b1), n1, b1.shape[-1], _lib.ptr(row_offsets), G, int(max_rows),
                                               _lib.ptr(b2), A, b2.shape[-1], gs, version, _lib.ptr(ious),
                                               _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_box_iou_rotated_grouped_f32")
    return ious
