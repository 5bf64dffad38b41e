"""The control path of the Oriented R-CNN heads in a handful of launches (csrc/orpn.hip) -- host side.

Reference: /root/reference/python/jdet/models/roi_heads/oriented_rpn_head.py:135-222 (_get_bboxes_single),
models/boxes/coder.py:372-433 (MidpointOffsetCoder.decode), ops/bbox_transforms.py:501-671, models/boxes/sampler.py:57-180.

As tensor operations the proposal stage is ~280 launches per image and every sampler call ~70 (two top-k's of 6e5 values, an
argsort, a dozen gathers): 1 400 launches and 7 ms of the 60 ms Oriented R-CNN / VAN-B3 step, none of them more than a few
microseconds of work.  Here the proposals of a whole batch are 12 launches (an exact radix select of the nms_pre best scores of
every (image, level), one workgroup per image that decodes, masks, sorts 10 000 keys in LDS and writes the NMS input, the NMS,
a finishing pass) and a sampler call is 6.  The tensor forms stay: they are the CPU path, the route for inputs the kernels do
not take, and what tests/test_gpu_orpn.py compares against (``_ON = False``)."""
import ctypes

import torch

from .. import _lib

_ON = True      # False: the tensor-operation routes (what these kernels are tested against)


def _f6(vals, default):
    vals = tuple(vals) if vals is not None else (default,) * 6
    assert len(vals) == 6
    return (ctypes.c_float * 6)(*[float(v) for v in vals])


# ---- RandomSampler.sample_masked -------------------------------------------------------------------------------------
def sampler_applies(gt_inds, pri, num):
    return (_ON and gt_inds.is_cuda and gt_inds.dtype == torch.int32 and gt_inds.dim() == 1 and pri.is_cuda
            and pri.dtype in (torch.float32, torch.float64) and 0 < num <= 1024)


def sample_masked(gt_inds, valid, k_gt, pri, num, num_pos, neg_pos_ub):
    """(inds, is_pos, valid, assigned, counts) of include/rsdet.h: rsdet_sample_masked; gt_inds (n_props,) int32, valid
    (n_props,) bool or None, pri (k_gt + n_props,) float32 / float64."""
    lib = _lib.load()
    dev = gt_inds.device
    gt_inds, pri = gt_inds.contiguous(), pri.contiguous()
    n_props = gt_inds.numel()
    assert pri.numel() == n_props + k_gt
    if valid is not None:
        valid = valid.contiguous()
        assert valid.dtype == torch.bool and valid.numel() == n_props
    inds = torch.empty((num,), dtype=torch.int64, device=dev)
    assigned = torch.empty((num,), dtype=torch.int64, device=dev)
    counts = torch.empty((2,), dtype=torch.int64, device=dev)
    flags = torch.empty((2, num), dtype=torch.bool, device=dev)
    ws_bytes = lib.rsdet_sample_masked_ws_size(num)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_sample_masked(_lib.ptr(gt_inds), _lib.ptr(valid), n_props, int(k_gt), _lib.ptr(pri),
                                 int(pri.dtype == torch.float64), int(num), int(num_pos), float(neg_pos_ub), _lib.ptr(inds),
                                 _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(assigned), _lib.ptr(counts), _lib.ptr(ws),
                                 ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_sample_masked")
    return inds, flags[0], flags[1], assigned, counts


# ---- MidpointOffsetCoder.decode / obb2hbb ----------------------------------------------------------------------------
def decode_applies(anchors, deltas):
    return (_ON and anchors.is_cuda and anchors.dtype == torch.float32 and deltas.dtype == torch.float32
            and anchors.dim() == 2 and deltas.dim() == 2 and anchors.shape[1] == 4 and deltas.shape[1] == 6)


def midpoint_offset_decode(anchors, deltas, means, stds, max_ratio):
    lib = _lib.load()
    anchors, deltas = anchors.contiguous(), deltas.contiguous()
    n = anchors.shape[0]
    out = torch.empty((n, 5), dtype=torch.float32, device=anchors.device)
    rc = lib.rsdet_midpoint_offset_decode_f32(_lib.ptr(anchors), _lib.ptr(deltas), n, _f6(means, 0.), _f6(stds, 1.),
                                              float(max_ratio), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rsdet_midpoint_offset_decode_f32")
    return out


def obb2hbb_applies(obb):
    return _ON and obb.is_cuda and obb.dtype == torch.float32 and obb.dim() == 2 and obb.shape[1] == 5


def obb2hbb(obb):
    lib = _lib.load()
    obb = obb.contiguous()
    out = torch.empty((obb.shape[0], 4), dtype=torch.float32, device=obb.device)
    _lib.check(lib.rsdet_obb2hbb_f32(_lib.ptr(obb), obb.shape[0], _lib.ptr(out), _lib.stream_ptr()), "rsdet_obb2hbb_f32")
    return out


# ---- the proposals of a batch ----------------------------------------------------------------------------------------
def _levels(scores, regs, anchors, nms_pre, nms_post, nms_thr, min_size, means, stds, max_ratio):
    d = _lib.OrpnLevels()
    N, A = scores[0].shape[0], scores[0].shape[3]
    d.n_img, d.n_levels, d.A, d.nms_pre, d.nms_post = N, len(scores), A, int(nms_pre), int(nms_post)
    d.nms_thr, d.min_size, d.max_ratio = float(nms_thr), float(min_size), float(max_ratio)
    d.means, d.stds = _f6(means, 0.), _f6(stds, 1.)
    for l, (s, r, a) in enumerate(zip(scores, regs, anchors)):
        d.hw[l] = s.shape[1] * s.shape[2]
        d.score[l], d.reg[l], d.anchors[l] = s.data_ptr(), r.data_ptr(), a.data_ptr()
    return d


def proposals_apply(cls_scores, bbox_preds, anchors, nms_pre, nms_post):
    if not (_ON and 0 < len(cls_scores) <= 7 and 0 < nms_pre <= 2048 and nms_post > 0):
        return False
    N, A = cls_scores[0].shape[:2]
    tot = 0
    for c, r, a in zip(cls_scores, bbox_preds, anchors):
        hw = c.shape[2] * c.shape[3]
        if not (c.is_cuda and c.dtype == torch.float32 and r.dtype == torch.float32 and a.dtype == torch.float32
                and c.shape[:2] == (N, A) and r.shape == (N, 6 * A) + tuple(c.shape[2:]) and a.shape == (hw * A, 4)
                and A * hw < (1 << 21)):
            return False
        tot += min(nms_pre, A * hw)
    return tot <= 16384


def pixel_major_sigmoid(cls):
    """sigmoid of an (N, A, H, W) map written as (N, H, W, A) -- the order of the reference's cls.permute(1, 2, 0).reshape(-1)
    -- in one launch."""
    N, A, H, W = cls.shape
    out = torch.empty((N, H, W, A), dtype=cls.dtype, device=cls.device)
    return torch.sigmoid(cls.permute(0, 2, 3, 1), out=out)


def proposals(scores, regs, anchors, nms_pre, nms_post, nms_thr, min_size, means, stds, max_ratio):
    """scores[l] (N, H, W, A): the SIGMOID of the classification maps, pixel-major (pixel_major_sigmoid); regs[l]
    (N, 6 A, H, W); anchors[l] (H W A, 4)
    -> (out (N, nms_post, 6), flags (N, nms_post) bool): include/rsdet.h, rsdet_orpn_proposals_f32."""
    lib = _lib.load()
    scores = [s.contiguous() for s in scores]
    regs = [r.contiguous() for r in regs]
    anchors = [a.contiguous() for a in anchors]
    d = _levels(scores, regs, anchors, nms_pre, nms_post, nms_thr, min_size, means, stds, max_ratio)
    ref = ctypes.byref(d)
    if not lib.rsdet_orpn_proposals_supported(ref):
        raise _lib.RsdetError("rsdet_orpn_proposals_f32: unsupported pyramid (see include/rsdet.h)")
    dev = scores[0].device
    N = scores[0].shape[0]
    out = torch.empty((N, int(nms_post), 6), dtype=torch.float32, device=dev)
    flags = torch.empty((N, int(nms_post)), dtype=torch.bool, device=dev)
    ws_bytes = lib.rsdet_orpn_proposals_ws_size(ref)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_orpn_proposals_f32(ref, _lib.ptr(out), _lib.ptr(flags), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_orpn_proposals_f32")
    return out, flags
