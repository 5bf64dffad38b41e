"""CPU oracle for the oriented-detection hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``rs_detection_amd`` never does: the product path
raises if ``librsdet_hip.so`` is missing instead of falling back to this.

Two layers:

* ``oracle.c``   -- ctypes bindings of ``librsdet_oracle.so`` (oracle/rsdet_oracle.cpp,
  this repo's own restatement, each function citing reference file:line);
* ``oracle.ref`` -- ctypes bindings of ``_ref/libjdet_ref.so`` = the reference's own
  embedded CPU sources compiled by ``build_ref.py`` (None when not built);
* ``oracle.ref_hip`` -- ctypes bindings of ``_ref/libjdet_ref_hip.so`` = the reference's CUDA-only
  kernels (DCN, RROIAlign, FeatureRefine, poly NMS) compiled AS DEVICE CODE for gfx950 by
  ``build_ref_hip.py`` (hipcc, no macro shim): takes torch CUDA tensors, GPU tests only;
* NumPy restatements of the reference's Jittor tensor code (box coder, anchor
  grid, AlignConv offsets, losses) in ``oracle.np_*`` functions below.

All citations are relative to /root/reference/python/jdet/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)
_U8 = ctypes.POINTER(ctypes.c_uint8)


def _fp(a):
    return a.ctypes.data_as(_F)


def _ip(a):
    return a.ctypes.data_as(_I)


def _up(a):
    return a.ctypes.data_as(_U8)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build(force=False):
    """Compile librsdet_oracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "librsdet_oracle.so")
    src = os.path.join(_HERE, "rsdet_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librsdet_oracle.so"])
    ref = os.path.join(_HERE, "_ref", "libjdet_ref.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref)):
        subprocess.check_call(["python3", os.path.join(_HERE, "build_ref.py")])
    ref_hip = os.path.join(_HERE, "_ref", "libjdet_ref_hip.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref_hip)):
        subprocess.check_call(["python3", os.path.join(_HERE, "build_ref_hip.py")])


class _COracle:
    def __init__(self):
        build()
        self.lib = ctypes.CDLL(os.path.join(_HERE, "librsdet_oracle.so"))

    # a1/a2 -- ops/box_iou_rotated.py:487-500, ops/box_iou_rotated_v1.py:492-505
    def box_iou_rotated(self, b1, b2, version=0):
        b1, b2 = _f32(b1), _f32(b2)
        n1, n2 = b1.shape[0], b2.shape[0]
        out = np.zeros((n1, n2), np.float32)
        if n1 and n2:
            assert b1.shape[1] == b2.shape[1] and b1.shape[1] in (5, 6)
            self.lib.oracle_box_iou_rotated(_fp(b1), n1, _fp(b2), n2, b1.shape[1], version, _fp(out))
        return out

    # a16 -- ops/nms_rotated.py:414-449 ; returns bool keep mask (n,)
    def nms_rotated(self, dets, order, thr):
        dets = _f32(dets)
        n, bl = dets.shape
        assert bl in (5, 6)
        order = np.ascontiguousarray(order, dtype=np.int32)
        keep = np.zeros(n, np.uint8)
        if n:
            self.lib.oracle_nms_rotated(_fp(dets), n, bl, _ip(order), ctypes.c_float(thr), _up(keep))
        return keep.astype(bool)

    # a12 -- ops/orn.py:17-43
    def arf_forward(self, weight, indices):
        weight = _f32(weight)
        indices = np.ascontiguousarray(indices, dtype=np.uint8)
        O, I, nOri, kH, kW = weight.shape
        nRot = indices.shape[3]
        out = np.zeros((O * nRot, I * nOri, kH, kW), np.float32)
        self.lib.oracle_arf_forward(_fp(weight), _up(indices), O, I, nOri, kH, kW, nRot, _fp(out))
        return out

    # ops/orn.py:45-72
    def arf_backward(self, indices, grad_out):
        indices = np.ascontiguousarray(indices, dtype=np.uint8)
        grad_out = _f32(grad_out)
        nOri, kH, kW, nRot = indices.shape
        O, I = grad_out.shape[0] // nRot, grad_out.shape[1] // nOri
        gw = np.zeros((O, I, nOri, kH, kW), np.float32)
        self.lib.oracle_arf_backward(_up(indices), _fp(grad_out), O, I, nOri, kH, kW, nRot, _fp(gw))
        return gw

    # 8(f)4 -- ops/orn.py:290-363: feature (N, nFeature*nOri) -> (direction (N, nFeature) uint8, aligned like feature)
    def rie_forward(self, feature, nOri):
        feature = _f32(feature)
        N, C = feature.shape[:2]
        F = C // nOri
        d, out = np.zeros((N, F), np.uint8), np.zeros((N, C), np.float32)
        self.lib.oracle_rie_forward(_fp(feature), N, F, nOri, _up(d), _fp(out))
        return d, out.reshape(feature.shape)

    def rie_backward(self, direction, grad_out, nOri):
        direction = np.ascontiguousarray(direction, dtype=np.uint8)
        grad_out = _f32(grad_out)
        N, F = direction.shape
        gi = np.zeros((N, F * nOri), np.float32)
        self.lib.oracle_rie_backward(_up(direction), _fp(grad_out), N, F, nOri, _fp(gi))
        return gi.reshape(grad_out.shape)

    @staticmethod
    def _geom(geom):
        return [int(geom[k]) for k in ("kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw")]

    @staticmethod
    def out_hw(H, W, kh, kw, ph, pw, sh, sw, dh, dw):
        return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1,
                (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)

    # a11 -- ops/dcn_v1.py:132-184 ; im (B,C,H,W), offset (B,dg*2*kh*kw,Ho,Wo) -> (C*kh*kw, B,Ho,Wo)
    def deform_im2col(self, im, offset, geom, dg=1):
        im, offset = _f32(im), _f32(offset)
        B, C, H, W = im.shape
        g = self._geom(geom)
        Ho, Wo = self.out_hw(H, W, *g)
        col = np.zeros((C * g[0] * g[1], B, Ho, Wo), np.float32)
        self.lib.oracle_deform_im2col(_fp(im), _fp(offset), C, H, W, *g, B, dg, _fp(col))
        return col

    # ops/dcn_v1.py:186-241
    def deform_col2im(self, col, offset, im_shape, geom, dg=1):
        col, offset = _f32(col), _f32(offset)
        B, C, H, W = im_shape
        g = self._geom(geom)
        gim = np.zeros((B, C, H, W), np.float32)
        self.lib.oracle_deform_col2im(_fp(col), _fp(offset), C, H, W, *g, B, dg, _fp(gim))
        return gim

    # ops/dcn_v1.py:244-306
    def deform_col2im_coord(self, col, im, offset, geom, dg=1):
        col, im, offset = _f32(col), _f32(im), _f32(offset)
        B, C, H, W = im.shape
        g = self._geom(geom)
        goff = np.zeros_like(offset)
        self.lib.oracle_deform_col2im_coord(_fp(col), _fp(im), _fp(offset), C, H, W, *g, B, dg, _fp(goff))
        return goff

    # a18 -- ops/roi_align_rotated_v1.py:71-147
    def rroi_align_v1_forward(self, feat, rois, out_hw, scale, sample_num, variant="v1"):
        """variant "v0": ops/roi_align_rotated.py:59-126 (f4)."""
        feat, rois = _f32(feat), _f32(rois)
        N, C, H, W = feat.shape
        R = rois.shape[0]
        PH, PW = out_hw
        out = np.zeros((R, C, PH, PW), np.float32)
        if R:
            getattr(self.lib, "oracle_rroi_align_%s_forward" % variant)(
                _fp(feat), _fp(rois), R, C, H, W, PH, PW, ctypes.c_float(scale), int(sample_num), _fp(out))
        return out

    # ops/roi_align_rotated_v1.py:193-298
    def rroi_align_v1_backward(self, grad_out, rois, feat_shape, scale, sample_num, variant="v1"):
        grad_out, rois = _f32(grad_out), _f32(rois)
        N, C, H, W = feat_shape
        R, _, PH, PW = grad_out.shape
        gf = np.zeros((N, C, H, W), np.float32)
        if R:
            getattr(self.lib, "oracle_rroi_align_%s_backward" % variant)(
                _fp(grad_out), _fp(rois), R, C, H, W, PH, PW, ctypes.c_float(scale), int(sample_num), _fp(gf))
        return gf

    # f4 -- ops/fr.py:113-232
    def feature_refine_forward(self, feat, boxes, scale, points=1):
        feat, boxes = _f32(feat), _f32(boxes)
        N, C, H, W = feat.shape
        assert boxes.size == N * H * W * 5 and points in (1, 5)
        out = np.zeros_like(feat)
        if feat.size:
            self.lib.oracle_feature_refine_forward(_fp(feat), _fp(boxes), N, C, H, W, ctypes.c_float(scale),
                                                   int(points), _fp(out))
        return out

    def feature_refine_backward(self, top, boxes, scale, points=1):
        top, boxes = _f32(top), _f32(boxes)
        N, C, H, W = top.shape
        gi = np.zeros_like(top)
        if top.size:
            self.lib.oracle_feature_refine_backward(_fp(top), _fp(boxes), N, C, H, W, ctypes.c_float(scale),
                                                    int(points), _fp(gi))
        return gi

    # f4 -- ops/convex_sort.py:93-154
    def convex_sort_scan(self, x, y, m, start, order, circular=True):
        x, y, m = _f32(x), _f32(y), _f32(m)
        nbs, npts = x.shape
        start = np.ascontiguousarray(start, dtype=np.int32).reshape(nbs)
        order = np.ascontiguousarray(order, dtype=np.int32)
        out = np.full((nbs, npts + 1 if circular else npts), -1, np.int32)
        if nbs and npts:
            self.lib.oracle_convex_sort_scan(_fp(x), _fp(y), _fp(m), _ip(start), _ip(order), nbs, npts,
                                             int(circular), _ip(out))
        return out

    # f4 -- ops/nms_poly.py:17-210 (fp32 in-model polygon NMS)
    def poly_iou_f32(self, p1, p2):
        p1, p2 = _f32(p1), _f32(p2)
        out = np.zeros((p1.shape[0], p2.shape[0]), np.float32)
        if out.size:
            self.lib.oracle_poly_iou_f32(_fp(p1), p1.shape[0], _fp(p2), p2.shape[0], _fp(out))
        return out

    def poly_nms(self, boxes, thr):
        """poly_nms (:186-210): kept original indices in descending-score order (stable argsort)."""
        boxes = _f32(boxes)
        order = np.argsort(-boxes[:, 8], kind="stable")
        d = np.ascontiguousarray(boxes[order])
        keep = np.zeros(len(d), np.uint8)
        if len(d):
            self.lib.oracle_poly_nms_sorted(_fp(d), len(d), ctypes.c_float(thr), _up(keep))
        return order[keep.astype(bool)]

    # a4 -- models/boxes/assigner.py:111-170
    def assign_wrt_overlaps(self, overlaps, pos_thr=0.5, neg_thr=0.4, min_pos_iou=0.0,
                            match_low_quality=True, gt_max_assign_all=True, gt_labels=None,
                            labels_filled=0):
        ov = _f32(overlaps)
        K, A = ov.shape
        neg_lo, neg_hi = (0.0, neg_thr) if not isinstance(neg_thr, tuple) else neg_thr
        gt_inds = np.zeros(A, np.int32)
        max_ov = np.zeros(A, np.float32)
        labels = np.zeros(A, np.int32) if gt_labels is not None else None
        gl = np.ascontiguousarray(gt_labels, dtype=np.int32) if gt_labels is not None else None
        self.lib.oracle_assign_wrt_overlaps(
            _fp(ov), K, A, ctypes.c_float(pos_thr), ctypes.c_float(neg_lo), ctypes.c_float(neg_hi),
            ctypes.c_float(min_pos_iou), int(match_low_quality), int(gt_max_assign_all),
            _ip(gl) if gl is not None else None, int(labels_filled), _ip(gt_inds), _fp(max_ov),
            _ip(labels) if labels is not None else None)
        return gt_inds, max_ov, labels


class _RefOracle:
    """The reference's own CPU sources (oracle/_ref).  ``available`` is False when absent."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libjdet_ref.so")
        self.available = os.path.exists(path)
        self.lib = ctypes.CDLL(path) if self.available else None

    def box_iou_rotated(self, b1, b2, version=0):
        b1, b2 = _f32(b1), _f32(b2)
        assert b1.shape[1] == 5 and b2.shape[1] == 5
        out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
        fn = self.lib.ref_box_iou_rotated if version == 0 else self.lib.ref_box_iou_rotated_v1
        if out.size:
            fn(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], _fp(out))
        return out

    def nms_rotated(self, dets, order, thr):
        dets = _f32(dets)
        n, bl = dets.shape
        order = np.ascontiguousarray(order, dtype=np.int32)
        keep = np.zeros(n, np.uint8)
        fn = self.lib.ref_nms_rotated5 if bl == 5 else self.lib.ref_nms_rotated6
        if n:
            fn(_fp(dets), n, _ip(order), ctypes.c_float(thr), _up(keep))
        return keep.astype(bool)

    def convex_sort_scan(self, x, y, m, start, order, circular=True):
        """The reference's own Graham-scan loop (ops/convex_sort.py:93-154) on prepared start / order arrays."""
        x, y, m = _f32(x), _f32(y), _f32(m)
        nbs, npts = x.shape
        start = np.ascontiguousarray(start, dtype=np.int32).reshape(nbs)
        order = np.ascontiguousarray(order, dtype=np.int32)
        out = np.full((nbs, npts + 1 if circular else npts), -1, np.int32)
        if nbs and npts:
            self.lib.ref_convex_sort(_fp(x), _fp(y), _fp(m), _ip(start), _ip(order), nbs, npts, int(circular),
                                     _ip(out))
        return out

    def arf_forward(self, weight, indices):
        weight = _f32(weight)
        indices = np.ascontiguousarray(indices, dtype=np.uint8)
        O, I, nOri, kH, kW = weight.shape
        nRot = indices.shape[3]
        assert O * I * nOri * kH * kW <= 65535, "reference CPU ARF overflows its uint16 index (SURVEY q3)"
        out = np.zeros((O * nRot, I * nOri, kH, kW), np.float32)
        self.lib.ref_arf_forward(_fp(weight), O, I, nOri, kH, kW, _up(indices), nRot, _fp(out))
        return out

    def arf_backward(self, indices, grad_out):
        indices = np.ascontiguousarray(indices, dtype=np.uint8)
        grad_out = _f32(grad_out)
        nOri, kH, kW, nRot = indices.shape
        O, I = grad_out.shape[0] // nRot, grad_out.shape[1] // nOri
        assert O * I * nOri * kH * kW <= 65535
        gw = np.zeros((O, I, nOri, kH, kW), np.float32)
        self.lib.ref_arf_backward(_up(indices), nOri, kH, kW, nRot, _fp(grad_out),
                                  grad_out.shape[0], grad_out.shape[1], _fp(gw))
        return gw

    def rie_forward(self, feature, nOri):
        feature = _f32(feature)
        N, C = feature.shape[:2]
        d, out = np.zeros((N, C // nOri), np.uint8), np.zeros((N, C), np.float32)
        self.lib.ref_rie_forward(_fp(feature), N, C, nOri, _up(d), _fp(out))
        return d, out.reshape(feature.shape)

    def rie_backward(self, direction, grad_out, nOri):
        direction = np.ascontiguousarray(direction, dtype=np.uint8)
        grad_out = _f32(grad_out)
        N, F = direction.shape
        gi = np.zeros((N, F * nOri), np.float32)
        self.lib.ref_rie_backward(_up(direction), N, F, nOri, _fp(grad_out), _fp(gi))
        return gi.reshape(grad_out.shape)


class _RefHipOracle:
    """The reference's CUDA-only kernels as gfx950 device code (oracle/build_ref_hip.py).  Arguments and results are
    torch CUDA tensors (float32, contiguous); every call runs on the null stream between two device synchronisations.
    ``available`` is False when the library was not built (no /root/reference at build time)."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libjdet_ref_hip.so")
        self.available = os.path.exists(path)
        self.lib = ctypes.CDLL(path) if self.available else None

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def _run(self, fn, *args):
        import torch
        torch.cuda.synchronize()
        rc = fn(*args)
        torch.cuda.synchronize()
        assert rc == 0, "reference HIP launch failed: %d" % rc

    @staticmethod
    def _g(geom):
        return [int(geom[k]) for k in ("kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw")]

    def deform_im2col(self, im, offset, geom, dg=1):
        """dcn_v1.py:309-339; im (B,C,H,W), offset (B,2*kh*kw*dg,Ho,Wo) -> columns (C*kh*kw, B, Ho, Wo)."""
        import torch
        B, C, H, W = im.shape
        Ho, Wo = offset.shape[2:]
        kh, kw = int(geom["kh"]), int(geom["kw"])
        col = torch.empty((C * kh * kw, B, Ho, Wo), dtype=torch.float32, device=im.device)
        self._run(self.lib.ref_hip_dcn_im2col, self._p(im.contiguous()), self._p(offset.contiguous()), C, H, W,
                  *self._g(geom), B, dg, self._p(col))
        return col

    def deform_col2im(self, col, offset, im_shape, geom, dg=1):
        """dcn_v1.py:376-410 -> grad_im (B,C,H,W)."""
        import torch
        B, C, H, W = im_shape
        gim = torch.empty((B, C, H, W), dtype=torch.float32, device=col.device)
        self._run(self.lib.ref_hip_dcn_col2im, self._p(col.contiguous()), self._p(offset.contiguous()), C, H, W,
                  *self._g(geom), B, dg, self._p(gim))
        return gim

    def deform_col2im_coord(self, col, im, offset, geom, dg=1):
        """dcn_v1.py:341-373 -> grad_offset, same shape as offset."""
        import torch
        B, C, H, W = im.shape
        goff = torch.empty_like(offset.contiguous())
        self._run(self.lib.ref_hip_dcn_col2im_coord, self._p(col.contiguous()), self._p(im.contiguous()),
                  self._p(offset.contiguous()), C, H, W, *self._g(geom), B, dg, self._p(goff))
        return goff

    def rroi_forward(self, feat, rois, out_hw, scale, sample_num):
        """roi_align_rotated_v1.py:300-325."""
        import torch
        N, C, H, W = feat.shape
        R = rois.shape[0]
        out = torch.empty((R, C, out_hw[0], out_hw[1]), dtype=torch.float32, device=feat.device)
        self._run(self.lib.ref_hip_rroi_forward, self._p(feat.contiguous()), self._p(rois.contiguous()), R, C, H, W,
                  out_hw[0], out_hw[1], ctypes.c_float(scale), ctypes.c_float(sample_num), self._p(out))
        return out

    def rroi_backward(self, grad_out, rois, feat_shape, scale, sample_num):
        """roi_align_rotated_v1.py:327-351."""
        import torch
        N, C, H, W = feat_shape
        R, _, PH, PW = grad_out.shape
        g = torch.empty((N, C, H, W), dtype=torch.float32, device=grad_out.device)
        self._run(self.lib.ref_hip_rroi_backward, self._p(grad_out.contiguous()), self._p(rois.contiguous()), R, N, C, H,
                  W, PH, PW, ctypes.c_float(scale), ctypes.c_float(sample_num), self._p(g))
        return g

    def fr_forward(self, feat, boxes, scale, points):
        """fr.py:234-240; boxes (N,H,W,5)."""
        import torch
        N, C, H, W = feat.shape
        out = torch.empty_like(feat.contiguous())
        self._run(self.lib.ref_hip_fr_forward, self._p(feat.contiguous()), self._p(boxes.contiguous()), N, C, H, W,
                  int(points), ctypes.c_float(scale), self._p(out))
        return out

    def fr_backward(self, top, boxes, scale, points):
        """fr.py:244-252."""
        import torch
        N, C, H, W = top.shape
        out = torch.empty_like(top.contiguous())
        self._run(self.lib.ref_hip_fr_backward, self._p(top.contiguous()), self._p(boxes.contiguous()), N, C, H, W,
                  int(points), ctypes.c_float(scale), self._p(out))
        return out

    def poly_nms_sorted(self, boxes_sorted, thr, contract=True):
        """nms_poly.py:197-229 on score-sorted (n, 9) boxes -> bool keep (n,) in the sorted order.  ``contract=False``:
        the build of the same text with floating-point contraction off."""
        n = boxes_sorted.shape[0]
        keep = np.zeros(n, np.uint8)
        fn = self.lib.ref_hip_poly_nms if contract else self.lib.ref_hip_nofma_poly_nms
        if n:
            self._run(fn, self._p(boxes_sorted.contiguous()), n, ctypes.c_float(thr), _up(keep))
        return keep.astype(bool)

    def poly_iou_pairs(self, p, q, contract=True):
        """devPolyIoU (nms_poly.py:135-150) of p[i], q[i] (n, 8) each."""
        import torch
        out = torch.empty((p.shape[0],), dtype=torch.float32, device=p.device)
        fn = self.lib.ref_hip_poly_iou_pairs if contract else self.lib.ref_hip_nofma_poly_iou_pairs
        if p.shape[0]:
            self._run(fn, self._p(p.contiguous()), self._p(q.contiguous()), p.shape[0], self._p(out))
        return out


_c = None
_ref = None
_ref_hip = None


def ref_hip():
    global _ref_hip
    if _ref_hip is None:
        build()
        _ref_hip = _RefHipOracle()
    return _ref_hip



def c():
    global _c
    if _c is None:
        _c = _COracle()
    return _c


def ref():
    global _ref
    if _ref is None:
        build()
        _ref = _RefOracle()
    return _ref


# ----------------------------------------------------------------------------
# NumPy restatements of the reference's tensor code (float32 arithmetic).
# ----------------------------------------------------------------------------
F32 = np.float32
PI32 = np.float32(np.pi)


def np_convex_sort_prepare(pts, masks):
    """The tensor code in front of the Graham scan (ops/convex_sort.py:67-84 == :159-176), fp32 NumPy:
    lowest valid point (first index on ties), cosine of every point's direction from it, descending order.
    Jittor's argmin / argsort tie rules are not pinned by anything in the reference: FIRST index and a STABLE
    descending sort are adopted (what torch.argmin / torch.argsort(stable=True) do)."""
    pts = _f32(pts)
    INF, EPS = np.float32(10000000), np.float32(0.000001)
    m = np.asarray(masks).astype(np.float32)
    x, y = np.ascontiguousarray(pts[:, :, 0]), np.ascontiguousarray(pts[:, :, 1])
    masked_y = m * y + (np.float32(1) - m) * INF
    nbs, npts = x.shape
    if npts == 0:
        return x, y, m, np.zeros((nbs,), np.int32), np.zeros((nbs, 0), np.int32)
    start = masked_y.argmin(1).astype(np.int32)
    sx = np.take_along_axis(x, start[:, None].astype(np.int64), 1)
    sy = np.take_along_axis(y, start[:, None].astype(np.int64), 1)
    cos = (x - sx) / np.sqrt((x - sx) * (x - sx) + (y - sy) * (y - sy) + EPS)
    order = np.argsort(-cos, axis=1, kind="stable").astype(np.int32)
    return x, y, m, start, order


def np_convex_sort(pts, masks, circular=True, scan=None):
    """convex_sort (ops/convex_sort.py:196-201) = prepare + scan; `scan` defaults to the C restatement."""
    x, y, m, start, order = np_convex_sort_prepare(pts, masks)
    return (scan or c().convex_sort_scan)(x, y, m, start, order, circular)


def np_norm_angle(angle, version="le135"):
    """models/boxes/box_ops.py:176-182 (Python-style float mod)."""
    lo = F32(-np.pi / 2) if version == "le90" else F32(-np.pi / 4)
    period = F32(np.pi)
    return np.mod(angle.astype(F32) - lo, period).astype(F32) + lo


def np_bbox2delta_rotated(proposals, gt, means=(0,) * 5, stds=(1,) * 5):
    """models/boxes/box_ops.py:184-230."""
    p, g = proposals.astype(F32), gt.astype(F32)
    cosa, sina = np.cos(p[..., 4]), np.sin(p[..., 4])
    cx, cy = g[..., 0] - p[..., 0], g[..., 1] - p[..., 1]
    dx = (cosa * cx + sina * cy) / p[..., 2]
    dy = (-sina * cx + cosa * cy) / p[..., 3]
    dw = np.log(g[..., 2] / p[..., 2])
    dh = np.log(g[..., 3] / p[..., 3])
    da = np_norm_angle(g[..., 4] - p[..., 4]) / PI32
    d = np.stack([dx, dy, dw, dh, da], -1).astype(F32)
    return ((d - np.asarray(means, F32)) / np.asarray(stds, F32)).astype(F32)


def np_delta2bbox_rotated(rois, deltas, means=(0,) * 5, stds=(1,) * 5, wh_ratio_clip=16 / 1000):
    """models/boxes/box_ops.py:233-289 (single class: deltas (N,5))."""
    r = rois.astype(F32)
    d = (deltas.astype(F32) * np.asarray(stds, F32) + np.asarray(means, F32)).astype(F32)
    max_ratio = F32(np.abs(np.log(wh_ratio_clip)))
    dw = np.clip(d[:, 2], -max_ratio, max_ratio)
    dh = np.clip(d[:, 3], -max_ratio, max_ratio)
    c, s = np.cos(r[:, 4]), np.sin(r[:, 4])
    gx = d[:, 0] * r[:, 2] * c - d[:, 1] * r[:, 3] * s + r[:, 0]
    gy = d[:, 0] * r[:, 2] * s + d[:, 1] * r[:, 3] * c + r[:, 1]
    gw = r[:, 2] * np.exp(dw)
    gh = r[:, 3] * np.exp(dh)
    ga = np_norm_angle(PI32 * d[:, 4] + r[:, 4])
    return np.stack([gx, gy, gw, gh, ga], -1).astype(F32)


def np_s2anet_grid_anchors(featmap_size, stride, scale=4, ratio=1.0):
    """models/boxes/anchor_generator.py:22-78 (one base anchor, x fastest)."""
    fh, fw = featmap_size
    base = F32(stride)
    ctr = F32(0.5) * (base - 1)
    hr = np.sqrt(F32(ratio))
    w = base * (F32(1) / hr) * F32(scale)
    h = base * hr * F32(scale)
    xs = (np.arange(fw, dtype=F32) * F32(stride))
    ys = (np.arange(fh, dtype=F32) * F32(stride))
    xx = np.tile(xs, fh)
    yy = np.repeat(ys, fw)
    a = np.zeros((fh * fw, 5), F32)
    a[:, 0] = ctr + xx
    a[:, 1] = ctr + yy
    a[:, 2] = w
    a[:, 3] = h
    return a


def np_align_conv_offset(anchors, featmap_size, stride, ks=3):
    """models/roi_heads/s2anet_head.py:676-713 -> (2*ks*ks, H, W), (y,x) per tap."""
    fh, fw = featmap_size
    pad = (ks - 1) // 2
    idx = np.arange(-pad, pad + 1, dtype=F32)
    yy, xx = np.meshgrid(idx, idx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    yc, xc = np.meshgrid(np.arange(fh, dtype=F32), np.arange(fw, dtype=F32), indexing="ij")
    xc, yc = xc.reshape(-1), yc.reshape(-1)
    x_conv, y_conv = xc[:, None] + xx, yc[:, None] + yy
    a = anchors.astype(F32)
    s = F32(stride)
    x_ctr, y_ctr, w, h = a[:, 0] / s, a[:, 1] / s, a[:, 2] / s, a[:, 3] / s
    cos, sin = np.cos(a[:, 4]), np.sin(a[:, 4])
    dw, dh = w / F32(ks), h / F32(ks)
    x, y = dw[:, None] * xx, dh[:, None] * yy
    xr = cos[:, None] * x - sin[:, None] * y
    yr = sin[:, None] * x + cos[:, None] * y
    off_x = xr + x_ctr[:, None] - x_conv
    off_y = yr + y_ctr[:, None] - y_conv
    off = np.stack([off_y, off_x], -1).astype(F32)
    return off.reshape(a.shape[0], -1).T.reshape(-1, fh, fw).copy()


def np_rotated_box_to_poly(r):
    """models/boxes/box_ops.py:633-654: corners tl,tr,br,bl of the unrotated rect."""
    r = r.astype(F32)
    n = r.shape[0]
    if n == 0:
        return np.zeros((0, 8), F32)
    hw, hh = r[:, 2] / 2, r[:, 3] / 2
    xs = np.stack([-hw, hw, hw, -hw], 1)
    ys = np.stack([-hh, -hh, hh, hh], 1)
    c, s = np.cos(r[:, 4])[:, None], np.sin(r[:, 4])[:, None]
    px = c * xs - s * ys + r[:, 0:1]
    py = s * xs + c * ys + r[:, 1:2]
    return np.stack([px, py], -1).reshape(n, 8).astype(F32)


def np_sigmoid_focal_loss(pred, target_1based, weight, gamma=2.0, alpha=0.25, avg_factor=None):
    """models/losses/focal_loss.py:5-34,36-96 (labels 1-based, 0 = background)."""
    pred = pred.astype(np.float64)
    C = pred.shape[1]
    t = (np.arange(1, C + 1)[None, :] == target_1based[:, None]).astype(np.float64)
    p = 1 / (1 + np.exp(-pred))
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt ** gamma
    bce = np.maximum(pred, 0) - pred * t + np.log1p(np.exp(-np.abs(pred)))
    loss = bce * fw
    if weight is not None:
        loss = loss * weight.reshape(-1, 1)
    return loss.sum() / avg_factor if avg_factor is not None else loss.mean()


def np_smooth_l1_loss(pred, target, weight, beta=1.0 / 9, avg_factor=None):
    """models/losses/smooth_l1_loss.py:5-54."""
    d = np.abs(pred.astype(np.float64) - target.astype(np.float64))
    loss = np.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    if weight is not None:
        loss = loss * weight
    return loss.sum() / avg_factor if avg_factor is not None else loss.mean()
