"""Polygon IoU and the tile-merge polygon NMS on the GPU (csrc/poly_iou.hip).

Reference: ``iou_poly`` (/root/reference/python/jdet/ops/nms_poly.py:247-252, shapely) and
``py_cpu_nms_poly_fast`` (/root/reference/python/jdet/data/devkits/result_merge.py:66-126).  Quadrilaterals are
(n, 8) float64 ``x1,y1,...,x4,y4``; results are float64.  shapely is absent here: parity unpinned."""
import numpy as np
import torch

from rs_detection_amd import _lib


def _f64_cuda(a, device=None):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    a = a.to(dtype=torch.float64)
    if not a.is_cuda:
        if device is None:
            raise _lib.RsdetError("polygon ops run on the GPU only; pass CUDA tensors or a device (no CPU fallback)")
        a = a.to(device)
    return a.contiguous()


def poly_iou_matrix(polys1, polys2, device=None):
    """(n1,8) x (n2,8) -> (n1,n2) float64 IoU (denominator max(union, 0.01), nms_poly.py:251)."""
    p1, p2 = _f64_cuda(polys1, device), _f64_cuda(polys2, device)
    assert p1.dim() == 2 and p1.shape[1] == 8 and p2.dim() == 2 and p2.shape[1] == 8
    out = torch.empty((p1.shape[0], p2.shape[0]), dtype=torch.float64, device=p1.device)
    if out.numel():
        rc = _lib.load().rsdet_poly_iou_f64(_lib.ptr(p1), p1.shape[0], _lib.ptr(p2), p2.shape[0], _lib.ptr(out),
                                            _lib.stream_ptr())
        _lib.check(rc, "rsdet_poly_iou_f64")
    return out


def iou_poly(poly1, poly2, device=None):
    """nms_poly.py:247-252 for one pair (8,), (8,)."""
    a = _f64_cuda(np.asarray(poly1, np.float64).reshape(1, 8) if not torch.is_tensor(poly1) else poly1.reshape(1, 8), device)
    b = _f64_cuda(np.asarray(poly2, np.float64).reshape(1, 8) if not torch.is_tensor(poly2) else poly2.reshape(1, 8), device)
    return float(poly_iou_matrix(a, b)[0, 0])


def nms_poly(dets, thresh, device=None):
    """py_cpu_nms_poly_fast (result_merge.py:66-126): dets (n, 9) = 8 polygon coordinates + score.  Returns the
    kept indices in descending-score order, like the reference's ``keep`` list.  Ties in the score are broken by
    index (stable sort); the reference's ``argsort()[::-1]`` leaves them to NumPy's unstable quicksort."""
    d = _f64_cuda(dets, device)
    n = d.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=d.device)
    order = torch.argsort(d[:, 8], descending=True, stable=True)
    polys = d[order, :8].contiguous()
    lib = _lib.load()
    keep = torch.empty((n,), dtype=torch.uint8, device=d.device)
    ws_bytes = lib.rsdet_nms_hbb_ws_size(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=d.device)
    rc = lib.rsdet_nms_poly_sorted_f64(_lib.ptr(polys), n, float(thresh), _lib.ptr(keep), _lib.ptr(ws), ws_bytes,
                                       _lib.stream_ptr())
    _lib.check(rc, "rsdet_nms_poly_sorted_f64")
    return order[keep.bool()]


# ---- in-model fp32 polygon NMS: ops/nms_poly.py:186-224 ------------------------------------
def poly_iou_f32(polys1, polys2):
    """devPolyIoU (nms_poly.py:100-132) for every pair: (n1,8) x (n2,8) float32 CUDA -> (n1,n2) float32."""
    _lib.require_cuda_f32(polys1, polys2)
    p1, p2 = polys1.contiguous(), polys2.contiguous()
    assert p1.dim() == 2 and p1.shape[1] == 8 and p2.dim() == 2 and p2.shape[1] == 8
    out = torch.empty((p1.shape[0], p2.shape[0]), dtype=torch.float32, device=p1.device)
    if out.numel():
        rc = _lib.load().rsdet_poly_iou_f32(_lib.ptr(p1), p1.shape[0], _lib.ptr(p2), p2.shape[0], _lib.ptr(out),
                                            _lib.stream_ptr())
        _lib.check(rc, "rsdet_poly_iou_f32")
    return out


def poly_nms(boxes, nms_overlap_thresh):
    """nms_poly.py:186-210: boxes (n, 9) = 8 polygon coordinates + score, float32 CUDA.  Returns the kept ORIGINAL
    indices in descending-score order (`order_t[keep]`, :210).  Equal scores keep their index order (stable sort;
    Jittor's argsort tie rule is not pinned by the reference)."""
    assert boxes.dim() == 2 and boxes.shape[1] == 9  # :187
    _lib.require_cuda_f32(boxes)                      # :188 assert jt.flags.use_cuda
    n = boxes.shape[0]
    order = torch.argsort(boxes[:, 8], descending=True, stable=True)
    if n == 0:
        return order
    boxes_sorted = boxes[order].contiguous()
    lib = _lib.load()
    keep = torch.empty((n,), dtype=torch.uint8, device=boxes.device)
    ws_bytes = lib.rsdet_nms_hbb_ws_size(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=boxes.device)
    rc = lib.rsdet_poly_nms_sorted_f32(_lib.ptr(boxes_sorted), n, float(nms_overlap_thresh), _lib.ptr(keep),
                                       _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_poly_nms_sorted_f32")
    return order[keep.bool()]


def multiclass_poly_nms(bboxes, scores, labels, thresh):
    """nms_poly.py:212-224: class-aware by shifting every class onto its own coordinate range.  bboxes (n, 8),
    scores (n,), labels (n,) -> dets (k, 9), labels (k,) in descending-score order."""
    if bboxes.shape[0] == 0:
        return torch.cat([bboxes, scores[:, None]], dim=1), labels
    max_coordinate = bboxes.max() - bboxes.min()
    offsets = labels.to(bboxes.dtype) * (max_coordinate + 1)
    bboxes_for_nms = bboxes + offsets[:, None]
    keep = poly_nms(torch.cat([bboxes_for_nms, scores[:, None]], dim=1), thresh)
    bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
    return torch.cat([bboxes, scores[:, None]], dim=1), labels
