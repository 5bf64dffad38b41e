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
