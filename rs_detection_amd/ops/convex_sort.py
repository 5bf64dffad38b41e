"""jdet.ops.convex_sort on MI355X.

Mirror of /root/reference/python/jdet/ops/convex_sort.py:196-201; one fused kernel (csrc/convex_sort.hip) instead of
the reference's argmin / gather / sqrt / argsort tensor ops plus the scan kernel.  No CPU fallback.
"""
import torch

from .. import _lib

__all__ = ["convex_sort"]


def convex_sort(pts, masks, circular=True):
    """pts (nbs, npts, 2) float32, masks (nbs, npts) bool / 0-1 -> (nbs, npts + 1 if circular else npts) int32 hull
    indices in scan order, -1 in unused slots (no gradient: the reference returns an index tensor)."""
    assert pts.size(0) == masks.size(0) and pts.size(1) == masks.size(1)  # :197
    _lib.require_cuda_f32(pts)
    if not masks.is_cuda:
        raise _lib.RsdetError("rs_detection_amd ops run on the GPU only (masks on %s); no CPU fallback" % masks.device)
    lib = _lib.load()
    nbs, npts = pts.shape[0], pts.shape[1]
    pts = pts.detach().contiguous()
    m = masks.detach().to(torch.float32).contiguous()  # :163 masks.cast(pts.dtype)
    out = torch.empty((nbs, npts + 1 if circular else npts), dtype=torch.int32, device=pts.device)
    ws_bytes = lib.rsdet_convex_sort_ws_size(nbs, npts)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=pts.device) if ws_bytes else None
    rc = lib.rsdet_convex_sort_f32(_lib.ptr(pts), _lib.ptr(m), nbs, npts, int(bool(circular)), _lib.ptr(out),
                                   _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_convex_sort_f32")
    return out
