"""Horizontal-box NMS = the role of Jittor's built-in ``jt.nms`` (third-party; call sites
/root/reference/python/jdet/ops/nms.py:9,44 and models/roi_heads/oriented_rpn_head.py:219).

``nms(dets (n,5)=[x1,y1,x2,y2,score], thresh)`` -> kept indices in descending-score order.  Jittor's source is
not vendored in the reference: PARITY UNPINNED; adopted semantics: stable descending sort, "+1" pixel-convention
IoU, suppression on IoU > thresh.  Kernel: csrc/nms_rotated.hip (hbb mask + the shared device sweep)."""
import torch

from .. import _lib

__all__ = ["nms"]


def nms(dets, thresh, plus_one=True):
    n = dets.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dets.device)
    _lib.require_cuda_f32(dets)
    lib = _lib.load()
    order = torch.argsort(dets[:, 4], descending=True, stable=True)
    boxes = dets[order, :4].contiguous()
    keep = torch.empty((n,), dtype=torch.uint8, device=dets.device)
    ws_bytes = lib.rsdet_nms_hbb_ws_size(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    rc = lib.rsdet_nms_hbb_sorted_f32(_lib.ptr(boxes), n, float(thresh), int(bool(plus_one)), _lib.ptr(keep),
                                      _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_nms_hbb_sorted_f32")
    return order[keep.bool()]
