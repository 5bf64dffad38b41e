"""jdet.ops.roi_align_rotated_v1 on MI355X: ROIAlignRotated_v1.

Mirror of /root/reference/python/jdet/ops/roi_align_rotated_v1.py:300-373;
kernels in csrc/rroi_align.hip.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .layout import nhwc_to_nchw, transpose_last2

_NCHW_GATHER = True     # backward writes NCHW directly from the tiled gather (C % 4 == 0); False: channels-last gather + a layout turn

__all__ = ["ROIAlignRotated_v1", "roi_align_rotated_v1", "rroi_align"]


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _RotatedROIAlign_v1(torch.autograd.Function):
    """`variant` = "v1" (this module's reference) or "v0" (ops/roi_align_rotated.py, see roi_align_rotated.py here):
    the two share kernels and differ in the RoI frame only."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio, variant="v1"):
        assert rois.shape[1] == 6  # :306
        assert variant in ("v0", "v1")
        _lib.require_cuda_f32(input, rois)
        lib = _lib.load()
        input, rois = input.contiguous(), rois.contiguous()
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(input.shape), output_size, float(spatial_scale), int(sampling_ratio), variant)
        N, C, H, W = input.shape
        R = rois.shape[0]
        out = torch.empty((R, C, output_size[0], output_size[1]), dtype=input.dtype, device=input.device)
        name = "rsdet_rroi_align_%s_forward_f32" % variant
        rc = getattr(lib, name)(_lib.ptr(input), _lib.ptr(rois), R, C, H, W, output_size[0], output_size[1],
                                float(spatial_scale), int(sampling_ratio), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, name)
        ctx.index = None
        if _INDEX_AT_FORWARD and ctx.needs_input_grad[0] and sampling_ratio > 0 and R > 0 and C % 4 == 0 \
                and not torch.cuda.is_current_stream_capturing():
            ctx.index = _build_index(lib, rois, N, H, W, output_size, float(spatial_scale), int(sampling_ratio), variant)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        shape, output_size, scale, sr, variant = ctx.cfg
        return (rroi_align_backward(grad_output, rois, shape, output_size, scale, sr, variant, index=ctx.index),
                None, None, None, None, None)


# True: the backward's inverted index is built at forward time on a side stream (_build_index).  Measured on the Oriented
# R-CNN / VAN-B3 step (round 6, same box, two runs each): 51.9 ms with the index built in the backward call, 53.7 - 53.8 ms
# with the side stream -- the cross-stream edges and the kernels that now run beside the forward cost more than the ~55 us
# per call they take out of the backward.  Off; the split entry points stay (tests/test_gpu_ops.py pins them).
_INDEX_AT_FORWARD = False
_SIDE = {}                   # device index -> the side stream the indices are built on


def _build_index(lib, rois, N, H, W, output_size, scale, sr, variant):
    """The inverted index of the gather-form backward (pixel -> (RoI, bin) rows and weights: a function of the RoIs and the
    geometry only), built NOW on a side stream beside the forward kernel instead of in front of the backward's gather,
    where its four small launches (count, scan, fill + a clear: ~55 us) were 40 % of the backward call.  Returns
    (workspace, event recorded behind the build)."""
    dev = rois.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE.get(dev.index)
    if side is None:
        side = _SIDE[dev.index] = torch.cuda.Stream(dev)
    R, PH, PW = rois.shape[0], output_size[0], output_size[1]
    ws_bytes = lib.rsdet_rroi_align_v1_backward_gather_ws_size(R, PH, PW, sr, N, H, W)
    side.wait_stream(main)                                  # the RoIs are ready
    with torch.cuda.stream(side):
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        name = "rsdet_rroi_align_%s_backward_index_f32" % variant
        rc = getattr(lib, name)(_lib.ptr(rois), R, N, H, W, PH, PW, scale, sr, _lib.ptr(ws), ws_bytes,
                                ctypes.c_void_p(side.cuda_stream))
        _lib.check(rc, name)
        ev = torch.cuda.Event()
        ev.record(side)
    rois.record_stream(side)
    return ws, ws_bytes, ev


def rroi_align_backward(grad_output, rois, shape, output_size, scale, sr, variant="v1", index=None, rows=None):
    """grad_feat (N,C,H,W) of _RotatedROIAlign_v1 (roi_align_rotated_v1.py:329-351); a plain function so that the bench
    can replay it from a hipGraph (device time, like the forward rows).  ``index``: what _build_index left at forward time;
    ``rows``: the channels-last gradient rows (R, PH*PW, C) when the caller already has them (the levels of one extractor
    share them)."""
    lib = _lib.load()
    N, C, H, W = shape
    go = grad_output.contiguous()
    R, PH, PW = rois.shape[0], output_size[0], output_size[1]
    if index is not None and sr > 0 and R > 0 and C % 4 == 0:
        ws, ws_bytes, ev = index
        main = torch.cuda.current_stream(go.device)
        main.wait_event(ev)
        ws.record_stream(main)
        go_t = transpose_last2(go.view(R, C, PH * PW))
        g = torch.empty((N, C, H, W), dtype=go.dtype, device=go.device)
        rc = lib.rsdet_rroi_align_backward_gather_indexed_f32(_lib.ptr(go_t), R, C, N, H, W, PH, PW, sr, 1, _lib.ptr(g),
                                                              _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, "rsdet_rroi_align_backward_gather_indexed_f32")
        return g
    if sr > 0 and R > 0:
        # gather form: no fp32 atomics (csrc/rroi_align.hip).  The gradient rows are turned channels-last once (small);
        # the result is written in NCHW directly by the tiled gather (rroi_gather_nchw_tile_kernel: 147 us for the call at
        # 2 x 256 x 256 x 256 with 512 RoIs) -- or, for C % 4 != 0 / _NCHW_GATHER = False, channels-last by the
        # one-wave-per-pixel gather and turned afterwards (169 us).
        nchw = _NCHW_GATHER and C % 4 == 0
        go_t = rows if rows is not None else transpose_last2(go.view(R, C, PH * PW))   # (R, 49, C): rows for the gather
        g = torch.empty((N, C, H, W) if nchw else (N, H, W, C), dtype=go.dtype, device=go.device)
        ws_bytes = lib.rsdet_rroi_align_v1_backward_gather_ws_size(R, PH, PW, sr, N, H, W)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=go.device)
        name = "rsdet_rroi_align_%s_backward_gather_%sf32" % (variant, "nchw_" if nchw else "")
        rc = getattr(lib, name)(_lib.ptr(go_t), _lib.ptr(rois), R, C, N, H, W, PH, PW, scale, sr,
                                _lib.ptr(g), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, name)
        return g if nchw else nhwc_to_nchw(g)
    grad_in = torch.zeros(shape, dtype=go.dtype, device=go.device)  # :345 memset
    name = "rsdet_rroi_align_%s_backward_f32" % variant
    rc = getattr(lib, name)(_lib.ptr(go), _lib.ptr(rois), rois.shape[0], C, H, W, output_size[0], output_size[1],
                            scale, sr, _lib.ptr(grad_in), _lib.stream_ptr())
    _lib.check(rc, name)
    return grad_in


_LEVELS_BACKWARD = True   # False: the per-level backward over all RoIs (what the one-index form is tested against)


class _RotatedROIAlignLevels(torch.autograd.Function):
    """OrientedSingleRoIExtractor's forward (oriented_single_level.py:91-114) as ONE launch: RoI n samples the map of its
    level ``lvls[n]`` (rsdet_rroi_align_v{0,1}_forward_levels_f32).  Backward: per level, the single-map backward over all
    RoIs with the other levels' RoIs moved outside the map -- exactly what the sync-free per-level forward did, so the
    gradients are the per-level path's."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, rois, lvls, output_size, scales, sampling_ratio, variant, *feats):
        assert rois.shape[1] == 6 and variant in ("v0", "v1") and 0 < len(feats) <= 8 and len(scales) == len(feats)
        lib = _lib.load()
        feats = [f.contiguous() for f in feats]
        rois = rois.contiguous()
        _lib.require_cuda_f32(rois, *feats)
        N, C = feats[0].shape[:2]
        assert all(f.shape[0] == N and f.shape[1] == C for f in feats)
        lv = _lib.RroiLevels()
        lv.n_levels = len(feats)
        for l, (f, sc) in enumerate(zip(feats, scales)):
            lv.feat[l], lv.H[l], lv.W[l], lv.scale[l] = _lib.ptr(f), f.shape[2], f.shape[3], float(sc)
        lvl32 = lvls.to(torch.int32).contiguous()
        R = rois.shape[0]
        out = torch.empty((R, C, output_size[0], output_size[1]), dtype=torch.float32, device=rois.device)
        name = "rsdet_rroi_align_%s_forward_levels_f32" % variant
        _lib.check(getattr(lib, name)(lv, _lib.ptr(rois), _lib.ptr(lvl32), R, C, output_size[0], output_size[1],
                                      int(sampling_ratio), _lib.ptr(out), _lib.stream_ptr()), name)
        ctx.save_for_backward(rois, lvl32)
        ctx.cfg = ([tuple(f.shape) for f in feats], tuple(output_size), tuple(float(s) for s in scales), int(sampling_ratio),
                   variant)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        rois, lvl = ctx.saved_tensors
        shapes, output_size, scales, sr, variant = ctx.cfg
        grads = []
        R, PH, PW = rois.shape[0], output_size[0], output_size[1]
        go = grad_output.contiguous()
        rows = transpose_last2(go.view(R, go.shape[1], PH * PW)) if sr > 0 and R > 0 else None   # once for all levels
        C = go.shape[1]
        if _LEVELS_BACKWARD and rows is not None and _NCHW_GATHER and C % 4 == 0 and go.dtype == torch.float32:
            # one inverted index over the levels' pixels (every RoI on its own level's geometry), a gather per level
            lib = _lib.load()
            N = shapes[0][0]
            lv = _lib.RroiLevels()
            lv.n_levels = len(shapes)
            outs = (ctypes.c_void_p * len(shapes))()
            for i, (shape, sc) in enumerate(zip(shapes, scales)):
                lv.H[i], lv.W[i], lv.scale[i] = shape[2], shape[3], sc
                g = torch.empty(shape, dtype=torch.float32, device=go.device) if ctx.needs_input_grad[6 + i] else None
                outs[i] = _lib.ptr(g)
                grads.append(g)
            ws_bytes = lib.rsdet_rroi_align_backward_levels_ws_size(lv, R, PH, PW, sr, N)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=go.device)
            name = "rsdet_rroi_align_%s_backward_levels_nchw_f32" % variant
            _lib.check(getattr(lib, name)(lv, outs, _lib.ptr(rows), _lib.ptr(rois), _lib.ptr(lvl), R, C, N, PH, PW, sr,
                                          _lib.ptr(ws), ws_bytes, _lib.stream_ptr()), name)
            return (None, None, None, None, None, None) + tuple(grads)
        for i, (shape, sc) in enumerate(zip(shapes, scales)):
            if not ctx.needs_input_grad[6 + i]:
                grads.append(None)
                continue
            on = (lvl == i)[:, None]                                              # (the per-level form's masked RoIs;
            r = rois.clone()                                                      #  scalars: no host-to-device copy)
            r[:, 1:3] = torch.where(on, rois[:, 1:3], -1e8)
            r[:, 3:5] = torch.where(on, rois[:, 3:5], 1.0)
            grads.append(rroi_align_backward(go, r, shape, output_size, sc, sr, variant, rows=rows))
        return (None, None, None, None, None, None) + tuple(grads)


def rroi_align_levels_applies(feats, rois):
    return (1 < len(feats) <= 8 and rois.is_cuda and rois.dim() == 2 and rois.shape[1] == 6 and rois.shape[0] > 0
            and all(f.is_cuda and f.dim() == 4 and f.dtype in (torch.float32, torch.bfloat16, torch.float16)
                    and f.shape[:2] == feats[0].shape[:2] for f in feats))


def rroi_align_levels(feats, rois, lvls, output_size, scales, sampling_ratio, variant="v1"):
    return _RotatedROIAlignLevels.apply(rois, lvls, tuple(output_size), tuple(scales), sampling_ratio, variant, *feats)


rroi_align = _RotatedROIAlign_v1.apply


def roi_align_rotated_v1(input, rois, output_size, spatial_scale, sampling_ratio):
    return rroi_align(input, rois, output_size, spatial_scale, sampling_ratio, "v1")


class ROIAlignRotated_v1(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=0):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align_rotated_v1(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
