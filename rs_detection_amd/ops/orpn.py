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


def sample_masked(gt_inds, valid, k_gt, pri, num, num_pos, neg_pos_ub, out=None):
    """(inds, is_pos, valid, assigned, counts) of include/rsdet.h: rsdet_sample_masked; gt_inds (n_props,) int32, valid
    (n_props,) bool or None, pri (k_gt + n_props,) float32 / float64.  ``out``: the five tensors to write (contiguous rows of
    the caller's batch)."""
    lib = _lib.load()
    dev = gt_inds.device
    gt_inds, pri = gt_inds.contiguous(), pri.contiguous()
    n_props = gt_inds.numel()
    assert pri.numel() == n_props + k_gt
    if valid is not None:
        valid = valid.contiguous()
        assert valid.dtype == torch.bool and valid.numel() == n_props
    if out is not None:
        inds, f0, f1, assigned, counts = out
        assert all(t.is_contiguous() for t in out) and inds.numel() == num and counts.numel() == 2
    else:
        inds = torch.empty((num,), dtype=torch.int64, device=dev)
        assigned = torch.empty((num,), dtype=torch.int64, device=dev)
        counts = torch.empty((2,), dtype=torch.int64, device=dev)
        flags = torch.empty((2, num), dtype=torch.bool, device=dev)
        f0, f1 = flags[0], flags[1]
    ws_bytes = lib.rsdet_sample_masked_ws_size(num)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_sample_masked(_lib.ptr(gt_inds), _lib.ptr(valid), n_props, int(k_gt), _lib.ptr(pri),
                                 int(pri.dtype == torch.float64), int(num), int(num_pos), float(neg_pos_ub), _lib.ptr(inds),
                                 _lib.ptr(f0), _lib.ptr(f1), _lib.ptr(assigned), _lib.ptr(counts), _lib.ptr(ws),
                                 ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_sample_masked")
    return inds, f0, f1, assigned, counts


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


# ---- the Oriented RPN's losses on the samples ------------------------------------------------------------------------
class RpnLossSpec:
    """Everything rsdet_orpn_loss needs besides the prediction maps (include/rsdet.h): the flat anchors and the inside-index
    list, per image the ground truth (K, 5) as the head sees it, the samples of rsdet_sample_masked stacked over the images
    (inds / is_pos / val / assigned (N, num), counts (N, 2)), the coder's constants and the losses' parameters."""
    __slots__ = ("anchors", "inside", "gts", "inds", "is_pos", "val", "assigned", "counts", "means", "stds", "beta", "w_cls",
                 "w_box", "pos_weight")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _loss_desc(spec, cls, reg):
    d = _lib.OrpnLoss()
    N, A = cls[0].shape[:2]
    d.n_img, d.n_levels, d.A, d.num = N, len(cls), A, spec.inds.shape[1]
    for l, (c, r) in enumerate(zip(cls, reg)):
        d.hw[l] = c.shape[2] * c.shape[3]
        d.cls[l], d.reg[l] = c.data_ptr(), r.data_ptr()
    return d


class _OrpnLoss(torch.autograd.Function):
    """forward(spec, cls_0 .. cls_{L-1}, reg_0 .. reg_{L-1}) -> 2 L scalars: loss_cls per level, then loss_bbox per level."""

    @staticmethod
    def forward(ctx, spec, *maps):
        lib = _lib.load()
        L = len(maps) // 2
        cls = [m.contiguous() for m in maps[:L]]
        reg = [m.contiguous() for m in maps[L:]]
        d = _loss_desc(spec, cls, reg)
        d.anchors, d.inside = spec.anchors.data_ptr(), (spec.inside.data_ptr() if spec.inside is not None else None)
        for b, g in enumerate(spec.gts):
            d.gt[b], d.k_gt[b] = g.data_ptr(), g.shape[0]
        d.inds, d.is_pos, d.val = spec.inds.data_ptr(), spec.is_pos.data_ptr(), spec.val.data_ptr()
        d.assigned, d.counts = spec.assigned.data_ptr(), spec.counts.data_ptr()
        d.means, d.stds = _f6(spec.means, 0.), _f6(spec.stds, 1.)
        d.beta, d.w_cls, d.w_box, d.pos_weight = float(spec.beta), float(spec.w_cls), float(spec.w_box), float(spec.pos_weight)
        dev = cls[0].device
        N, num = spec.inds.shape
        losses = torch.empty((2 * L,), dtype=torch.float32, device=dev)
        rec = torch.empty((lib.rsdet_orpn_loss_rec_floats(N, num),), dtype=torch.float32, device=dev)
        rc = lib.rsdet_orpn_loss_forward_f32(ctypes.byref(d), _lib.ptr(losses), _lib.ptr(rec), _lib.stream_ptr())
        _lib.check(rc, "rsdet_orpn_loss_forward_f32")
        ctx.save_for_backward(rec)
        ctx.shapes = [tuple(m.shape) for m in cls + reg]
        ctx.num = num
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        rec, = ctx.saved_tensors
        shapes = ctx.shapes
        L = len(shapes) // 2
        dev = rec.device
        zero = None
        gl = []
        for g in grads:
            if g is None:
                zero = torch.zeros((), dtype=torch.float32, device=dev) if zero is None else zero
                g = zero
            gl.append(g.reshape(()))
        g = torch.stack(gl).float()
        sizes = [s[0] * s[1] * s[2] * s[3] for s in shapes]
        arena = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)     # one fill for all ten gradient maps
        outs, at = [], 0
        for s, n in zip(shapes, sizes):
            outs.append(arena[at:at + n].view(s))
            at += n
        d = _loss_desc(_Num(ctx.num), outs[:L], outs[L:])
        rc = lib.rsdet_orpn_loss_backward_f32(ctypes.byref(d), _lib.ptr(rec), _lib.ptr(g), _lib.stream_ptr())
        _lib.check(rc, "rsdet_orpn_loss_backward_f32")
        return (None,) + tuple(outs)


class _Num:
    """(a spec that only carries the sample count, for the backward's descriptor)"""

    def __init__(self, num):
        self.inds = torch.empty((0, num))


def rpn_loss_applies(cls_scores, bbox_preds, n_img, num):
    if not (_ON and 0 < len(cls_scores) <= 8 and 0 < n_img <= 16 and 0 < num <= 1024):
        return False
    for c, r in zip(cls_scores, bbox_preds):
        if not (c.is_cuda and c.dtype == torch.float32 and r.dtype == torch.float32 and c.dim() == 4
                and r.shape == (c.shape[0], 6 * c.shape[1]) + tuple(c.shape[2:]) and r.numel() < (1 << 31)):
            return False
    return True


def rpn_loss(spec, cls_scores, bbox_preds):
    """-> (loss_cls per level, loss_bbox per level): two lists of 0-d tensors (rsdet_orpn_loss_forward_f32)."""
    L = len(cls_scores)
    out = _OrpnLoss.apply(spec, *cls_scores, *bbox_preds)
    return list(out[:L]), list(out[L:])


# ---- OrientedHead: sampled RoIs and targets --------------------------------------------------------------------------
def roi_targets_apply(props, gt, gt_labels):
    return (_ON and props.is_cuda and props.dtype == torch.float32 and props.dim() == 2 and props.shape[1] >= 5
            and props.is_contiguous() and gt.dtype == torch.float32 and gt.dim() == 2 and gt.shape[1] == 5
            and gt_labels.dtype == torch.int64)


def roi_targets(props, gt, gt_labels, sample, image, num_classes, means, stds, pos_weight, out):
    """include/rsdet.h: rsdet_orcnn_roi_targets_f32.  sample = (inds, is_pos, val, assigned) rows of one image; out = (rois,
    labels, label_weights, bbox_targets, bbox_weights) that image's rows of the batch (contiguous)."""
    lib = _lib.load()
    inds, is_pos, val, assigned = sample
    gt, gt_labels = gt.contiguous(), gt_labels.contiguous()
    assert all(t.is_contiguous() for t in out)
    rc = lib.rsdet_orcnn_roi_targets_f32(_lib.ptr(props), props.shape[1], props.shape[0], _lib.ptr(gt), _lib.ptr(gt_labels),
                                         gt.shape[0], _lib.ptr(inds), _lib.ptr(is_pos), _lib.ptr(val), _lib.ptr(assigned),
                                         inds.numel(), int(image), int(num_classes), _lib.host5(means, 0.), _lib.host5(stds, 1.),
                                         float(pos_weight), *[_lib.ptr(t) for t in out], _lib.stream_ptr())
    _lib.check(rc, "rsdet_orcnn_roi_targets_f32")


# ---- MaxIoUAssigner on horizontal boxes without the overlaps matrix ---------------------------------------------------
def hbb_assign_applies(bboxes, gt_bboxes):
    return (_ON and bboxes.is_cuda and bboxes.dtype == torch.float32 and gt_bboxes.dtype == torch.float32
            and bboxes.dim() == 2 and gt_bboxes.dim() == 2 and bboxes.shape[1] >= 4 and gt_bboxes.shape[1] >= 4
            and bboxes.stride(1) == 1 and gt_bboxes.stride(1) == 1 and 0 < gt_bboxes.shape[0] <= 1024 and bboxes.shape[0] > 0
            and not (bboxes.requires_grad or gt_bboxes.requires_grad))


def hbb_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality, gt_max_assign_all, eps=1e-6):
    """(gt_inds (A,) int32, max_overlaps (A,)) of include/rsdet.h: rsdet_hbb_assign_f32."""
    lib = _lib.load()
    A, K = bboxes.shape[0], gt_bboxes.shape[0]
    neg_lo, neg_hi = neg_iou_thr if isinstance(neg_iou_thr, (tuple, list)) else (0.0, neg_iou_thr)
    gt_inds = torch.empty((A,), dtype=torch.int32, device=bboxes.device)
    max_ov = torch.empty((A,), dtype=torch.float32, device=bboxes.device)
    nb = lib.rsdet_hbb_assign_ws_size(K)
    ws = torch.empty((nb,), dtype=torch.uint8, device=bboxes.device)
    rc = lib.rsdet_hbb_assign_f32(_lib.ptr(gt_bboxes), K, gt_bboxes.stride(0), _lib.ptr(bboxes), A, bboxes.stride(0), float(eps),
                                  float(pos_iou_thr), float(neg_lo), float(neg_hi), float(min_pos_iou),
                                  int(bool(match_low_quality)), int(bool(gt_max_assign_all)), _lib.ptr(gt_inds),
                                  _lib.ptr(max_ov), _lib.ptr(ws), nb, _lib.stream_ptr())
    _lib.check(rc, "rsdet_hbb_assign_f32")
    return gt_inds, max_ov
