"""jdet.ops.dcn_v1 on MI355X: DeformConv (deformable convolution v1, AlignConv's core).

Mirror of /root/reference/python/jdet/ops/dcn_v1.py:559-713.  Bilinear im2col /
col2im / col2im_coord are the hand-written kernels of csrc/deform_conv.hip; the
GEMMs (:447, :484, :547) go to rocBLAS/hipBLASLt through torch.matmul.

MI355X-first differences from the reference's host logic (free per SURVEY q17):
  * the column matrix of the forward pass is kept for backward (288 GB of HBM)
    instead of recomputing im2col for the weight gradient (:536-539);
  * the whole batch is one im2col "step" (the reference chunks by im2col_step
    only to bound memory); chunks are still honoured when a cap is set;
  * the offset gradient is computed only when the offset requires grad
    (AlignConv builds offsets under no_grad, s2anet_head.py:676 / SURVEY q16).
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from .layout import nchw_to_nhwc, nhwc_to_nchw

__all__ = ["DeformConv", "deform_conv", "deformable_im2col", "deformable_col2im", "deformable_col2im_coord",
           "deformable_im2col_nhwc", "deformable_col2im_nhwc"]


# bf16 columns + bf16 products under bf16 autocast (what autocast does to every other convolution of the step)
_LOWP_ALIGNCONV = True


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg):
    return _lib.DcnGeom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg)


def _out_hw(H, W, kh, kw, ph, pw, sh, sw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def deformable_im2col(im, offset, kernel, padding, stride, dilation, deformable_group=1, col_dtype=None):
    """dcn_v1.py:309-339: im (B,C,H,W), offset (B,dg*2*kh*kw,Ho,Wo) -> col (C*kh*kw, B*Ho*Wo).
    ``col_dtype=torch.bfloat16``: the columns are stored as bf16 (autocast step; 3x3 AlignConv geometry only)."""
    _lib.require_cuda_f32(im, offset)
    lib = _lib.load()
    im, offset = im.contiguous(), offset.contiguous()
    B, C, H, W = im.shape
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    Ho, Wo = _out_hw(H, W, kh, kw, ph, pw, sh, sw, dh, dw)
    assert tuple(offset.shape) == (B, deformable_group * 2 * kh * kw, Ho, Wo), "invalid offset shape"
    lowp = col_dtype == torch.bfloat16
    col = torch.empty((C * kh * kw, B * Ho * Wo), dtype=torch.bfloat16 if lowp else im.dtype, device=im.device)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_group)
    name = "rsdet_deform_im2col_bf16col_f32" if lowp else "rsdet_deform_im2col_f32"
    _lib.check(getattr(lib, name)(_lib.ptr(im), _lib.ptr(offset), g, _lib.ptr(col), _lib.stream_ptr()), name)
    return col


def deformable_col2im(col, offset, im_shape, kernel, padding, stride, dilation, deformable_group=1):
    """dcn_v1.py:376-410 -> grad_im (B,C,H,W)."""
    _lib.require_cuda_f32(col, offset)
    lib = _lib.load()
    col, offset = col.contiguous(), offset.contiguous()
    B, C, H, W = im_shape
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    grad_im = torch.zeros((B, C, H, W), dtype=col.dtype, device=col.device)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_group)
    _lib.check(lib.rsdet_deform_col2im_f32(_lib.ptr(col), _lib.ptr(offset), g, _lib.ptr(grad_im), _lib.stream_ptr()),
               "rsdet_deform_col2im_f32")
    return grad_im


def deformable_col2im_coord(col, im, offset, kernel, padding, stride, dilation, deformable_group=1):
    """dcn_v1.py:341-373 -> grad_offset like offset."""
    _lib.require_cuda_f32(col, im, offset)
    lib = _lib.load()
    col, im, offset = col.contiguous(), im.contiguous(), offset.contiguous()
    B, C, H, W = im.shape
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    grad_off = torch.empty_like(offset)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_group)
    _lib.check(lib.rsdet_deform_col2im_coord_f32(_lib.ptr(col), _lib.ptr(im), _lib.ptr(offset), g,
                                                 _lib.ptr(grad_off), _lib.stream_ptr()),
               "rsdet_deform_col2im_coord_f32")
    return grad_off


def deformable_im2col_nhwc(im_nhwc, offset, kernel, padding, stride, dilation, deformable_group=1):
    """Channels-last form: im (B,H,W,C) contiguous, offset (B,dg*2*kh*kw,Ho,Wo) -> colT (B*Ho*Wo, kh*kw*C)."""
    _lib.require_cuda_f32(im_nhwc, offset)
    lib = _lib.load()
    assert im_nhwc.is_contiguous()
    offset = offset.contiguous()
    B, H, W, C = im_nhwc.shape
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    Ho, Wo = _out_hw(H, W, kh, kw, ph, pw, sh, sw, dh, dw)
    assert tuple(offset.shape) == (B, deformable_group * 2 * kh * kw, Ho, Wo), "invalid offset shape"
    colT = torch.empty((B * Ho * Wo, kh * kw * C), dtype=im_nhwc.dtype, device=im_nhwc.device)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_group)
    _lib.check(lib.rsdet_deform_im2col_nhwc_f32(_lib.ptr(im_nhwc), _lib.ptr(offset), g, _lib.ptr(colT),
                                                _lib.stream_ptr()), "rsdet_deform_im2col_nhwc_f32")
    return colT


def deformable_col2im_nhwc(colT, offset, im_shape_nhwc, kernel, padding, stride, dilation, deformable_group=1):
    """colT (B*Ho*Wo, kh*kw*C) -> grad_im (B,H,W,C) (fp32 atomics, contiguous 256-B segments)."""
    _lib.require_cuda_f32(colT, offset)
    lib = _lib.load()
    colT, offset = colT.contiguous(), offset.contiguous()
    B, H, W, C = im_shape_nhwc
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    grad_im = torch.zeros((B, H, W, C), dtype=colT.dtype, device=colT.device)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_group)
    _lib.check(lib.rsdet_deform_col2im_nhwc_f32(_lib.ptr(colT), _lib.ptr(offset), g, _lib.ptr(grad_im),
                                                _lib.stream_ptr()), "rsdet_deform_col2im_nhwc_f32")
    return grad_im


class GatherIndexPlan:
    """The calls of ONE forward pass whose backward uses the gather-form col2im (the five AlignConv levels of a step):
    the first backward that needs its index builds the index of ALL of them -- pixels and items laid end to end, one
    histogram / scan / fill (csrc/deform_conv.hip, rsdet_deform_col2im_index_multi_f32: 5 launches instead of 5 per
    level) -- and every level then only gathers.  Offsets are inputs without gradient (s2anet_head.py:676), so all of
    them exist when the first backward runs."""
    MAX = 8     # RSDET_DCN_INDEX_MAX_LEVELS

    def __init__(self):
        self.calls = []          # (offset tensor (contiguous fp32), geometry tuple)
        self.built = {}          # chunk -> (ws, start byte offset, pix_base list, ent_row offset, ent_w offset)
        self.operands = {}       # (id(weight), version) -> (weight, prepared operand): one cast for the calls of the pass

    def add(self, offset, C, H, W, kernel, padding, stride, dilation, B):
        self.calls.append((offset, (C, H, W, *kernel, *padding, *stride, *dilation, B, 1)))
        return (self, len(self.calls) - 1)

    def index(self, i):
        """(start, ent_row, ent_w) device addresses of call i; builds its chunk of <= MAX calls on first use."""
        chunk = i // self.MAX
        if chunk not in self.built:
            lib = _lib.load()
            calls = self.calls[chunk * self.MAX:(chunk + 1) * self.MAX]
            lv = _lib.DcnIndexLevels()
            lv.n_levels = len(calls)
            npix = 0
            for l, (off, g) in enumerate(calls):
                lv.offset[l] = _lib.ptr(off)
                lv.geom[l] = _geom(*g)
                npix += g[11] * g[1] * g[2]
            ws_bytes = lib.rsdet_deform_col2im_index_multi_ws_size(lv)
            if ws_bytes == 0:
                raise RuntimeError("rsdet_deform_col2im_index_multi_ws_size: unsupported geometry")
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=calls[0][0].device)
            pix_base = (_lib.c_ll * (self.MAX + 1))()
            row_off, w_off = _lib.c_size_t(0), _lib.c_size_t(0)
            _lib.check(lib.rsdet_deform_col2im_index_multi_f32(lv, _lib.ptr(ws), ws_bytes, pix_base, ctypes.byref(row_off),
                                                               ctypes.byref(w_off), _lib.stream_ptr()),
                       "rsdet_deform_col2im_index_multi_f32")
            start_off = ((npix + 1) * 4 + 255) & ~255
            self.built[chunk] = (ws, start_off, list(pix_base), row_off.value, w_off.value)
        ws, start_off, pix_base, row_off, w_off = self.built[chunk]
        base = ws.data_ptr()
        return base + start_off + 4 * pix_base[i % self.MAX], base + row_off, base + w_off


_PLAN = None                    # the plan collecting the current forward pass (shared_gather_index), or None
_SHARED_INDEX = os.environ.get("RSDET_DCN_SHARED_INDEX", "1") != "0"


class shared_gather_index:
    """``with shared_gather_index():`` around the forward of several DeformConv calls: their gather-form col2im
    backwards share one index build (GatherIndexPlan)."""

    def __enter__(self):
        global _PLAN
        self.prev = _PLAN
        _PLAN = GatherIndexPlan() if _SHARED_INDEX else None
        return _PLAN

    def __exit__(self, *exc):
        global _PLAN
        _PLAN = self.prev
        return False


def _plan_slot(ctx, offset, C, H, W, kernel, padding, stride, dilation, B, dg):
    """Registers the call with the current plan (forward time); None when there is none or the gather form will not run."""
    if _PLAN is None or dg != 1 or not ctx.needs_input_grad[0]:
        return None
    return _PLAN.add(offset, C, H, W, kernel, padding, stride, dilation, B)


def deformable_col2im_gather_nhwc(colT, offset, im_shape_nhwc, kernel, padding, stride, dilation, slot=None,
                                  out_dtype=torch.float32):
    """colT (B*Ho*Wo, kh*kw*C) -> grad_im (B,H,W,C) without floating-point atomics (one deformable group):
    the scatter map is inverted on integers first, then every input pixel gathers its terms.  ``slot`` = the call's
    entry in a GatherIndexPlan: the index comes from the plan's shared build.  ``out_dtype`` bf16 (with bf16 columns and
    a slot): the sums, accumulated in fp32, are stored rounded -- the cast of a bf16 input's gradient without its pass."""
    lowp = colT.dtype == torch.bfloat16  # column gradient out of a bf16 GEMM (autocast step); grad_im stays fp32
    _lib.require_cuda_f32(None if lowp else colT, offset)
    lib = _lib.load()
    colT, offset = colT.contiguous(), offset.contiguous()
    B, H, W, C = im_shape_nhwc
    (kh, kw), (ph, pw), (sh, sw), (dh, dw) = kernel, padding, stride, dilation
    out_bf16 = out_dtype == torch.bfloat16 and lowp and slot is not None
    grad_im = torch.empty((B, H, W, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=colT.device)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, 1)
    if slot is not None:
        plan, i = slot
        start, ent_row, ent_w = plan.index(i)
        name = ("rsdet_deform_col2im_gather_indexed_nhwc_bf16col_bf16" if out_bf16 else
                "rsdet_deform_col2im_gather_indexed_nhwc_bf16col_f32" if lowp else "rsdet_deform_col2im_gather_indexed_nhwc_f32")
        _lib.check(getattr(lib, name)(_lib.ptr(colT), g, start, ent_row, ent_w, _lib.ptr(grad_im), _lib.stream_ptr()),
                   name)
        return grad_im
    ws_bytes = lib.rsdet_deform_col2im_gather_ws_size(g)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=colT.device)
    name = "rsdet_deform_col2im_gather_nhwc_bf16col_f32" if lowp else "rsdet_deform_col2im_gather_nhwc_f32"
    _lib.check(getattr(lib, name)(_lib.ptr(colT), _lib.ptr(offset), g, _lib.ptr(grad_im), _lib.ptr(ws), ws_bytes,
                                  _lib.stream_ptr()), name)
    return grad_im


class DeformConvFunctionNHWC(torch.autograd.Function):
    """groups == 1 fast path of DeformConvFunction (offset needs no gradient).

    Forward keeps the reference column layout (col (C*kh*kw, B*Ho*Wo), csrc deform_im2col) and
    writes each image's output with one GEMM straight into the NCHW result (no transposes).
    Backward is channels-last where it matters: gcolT[b] = gO[b]^T @ W(O, kh*kw*C) is a GEMM on
    strided views (no copies) and col2im runs the NHWC kernel, whose fp32 atomics are contiguous
    256-B segments per wave-instruction (7x faster than one-lane-per-row atomics at level 0).
    The forward column matrix is kept for the weight gradient instead of recomputed
    (dcn_v1.py:536-539; 288 GB of HBM make the 0.6 GB cheap)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, offset, weight, stride, padding, dilation, deformable_groups, lowp=False):
        ctx.cfg = (_pair(stride), _pair(padding), _pair(dilation), deformable_groups)
        B, C, H, W = input.shape
        O, _, kh, kw = weight.shape
        Ho, Wo = _out_hw(H, W, kh, kw, *ctx.cfg[1], *ctx.cfg[0], *ctx.cfg[2])
        # lowp (bf16 autocast step): bilinear sampling in fp32, columns stored as bf16, the three products on bf16 MFMA
        # with fp32 accumulation -- what autocast does to every other convolution of the step
        lowp = bool(lowp) and kh * kw == 9 and (C // deformable_groups) % 16 == 0 and W >= 2
        ctx.lowp = lowp
        cdt = torch.bfloat16 if lowp else input.dtype
        hw = Ho * Wo
        ctx.col_is_T = False
        geom = None
        if (not lowp) and _MFMA_ALIGNCONV and input.dtype == torch.float32:
            geom = _mfma_geom_f32(input, weight, *ctx.cfg)
        if geom is not None:
            # exact-fp32 implicit GEMM (csrc/alignconv_mfma.hip, v_mfma_f32_32x32x2_f32): one launch, the sampled columns
            # written once channels-last (bit-identical to the im2col kernel's) for the weight gradient
            lib = _lib.load()
            # a channels_last input (the fp32 step in channels_last) is consumed as it is and answered in kind: no
            # layout copies either way, and the backward below works on (positions, channels) views
            ctx.cl = not input.is_contiguous() and input.is_contiguous(memory_format=torch.channels_last)
            x_nhwc = nchw_to_nhwc(input)
            w_t = weight.permute(0, 2, 3, 1).reshape(O, kh * kw * C).contiguous()
            out = torch.empty((B, O, Ho, Wo), dtype=input.dtype, device=input.device,
                              memory_format=torch.channels_last if ctx.cl else torch.contiguous_format)
            colT = (torch.empty((B * hw, kh * kw * C), dtype=input.dtype, device=input.device)
                    if ctx.needs_input_grad[2] else None)
            off = offset.contiguous()
            _lib.check(lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_t), geom, O,
                                                        int(ctx.cl), _lib.ptr(out), _lib.ptr(colT), _lib.stream_ptr()),
                       "rsdet_alignconv_fwd_mfma_f32")
            ctx.col_is_T = True
            ctx.save_for_backward(off, weight, colT)
            ctx.in_shape = (B, C, H, W)
            ctx.slot = _plan_slot(ctx, off, C, H, W, (kh, kw), ctx.cfg[1], ctx.cfg[0], ctx.cfg[2], B, deformable_groups)
            return out
        col = deformable_im2col(input, offset, (kh, kw), ctx.cfg[1], ctx.cfg[0], ctx.cfg[2], deformable_groups,
                                col_dtype=cdt)
        w_flat = weight.reshape(O, C * kh * kw).to(cdt)
        # one strided-batched product over the images, straight into the NCHW result: col (K, B*hw) is viewed as
        # (B, K, hw) with strides (hw, B*hw, 1) -- no copies, one launch (and one library call) per level
        out = torch.empty((B, O, Ho, Wo), dtype=cdt, device=input.device)  # returned as is (callers apply ReLU in place)
        torch.bmm(w_flat.unsqueeze(0).expand(B, O, C * kh * kw), col.view(C * kh * kw, B, hw).permute(1, 0, 2),
                  out=out.view(B, O, hw))
        offset = offset.contiguous()
        ctx.save_for_backward(offset, weight, col)
        ctx.in_shape = (B, C, H, W)
        ctx.slot = _plan_slot(ctx, offset, C, H, W, (kh, kw), ctx.cfg[1], ctx.cfg[0], ctx.cfg[2], B, deformable_groups)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        offset, weight, col = ctx.saved_tensors
        stride, padding, dilation, dg = ctx.cfg
        B, C, H, W = ctx.in_shape
        O, _, kh, kw = weight.shape
        cdt = torch.bfloat16 if ctx.lowp else weight.dtype
        if getattr(ctx, "cl", False):
            return DeformConvFunctionNHWC._backward_channels_last(ctx, grad_output)
        go = grad_output.contiguous().to(cdt).view(B, O, -1)  # (B, O, Ho*Wo)
        hw = go.shape[2]
        grad_input = grad_weight = None
        # one (O, B*hw) copy of the output gradient (17 MB at level 0) turns both backward products into single
        # launches over the whole batch: measured 734 -> 630 us (data) and 771 -> 585 us (weight) at level 0
        go2 = go.transpose(0, 1).reshape(O, B * hw)
        if ctx.needs_input_grad[0]:
            w_ok = weight.permute(0, 2, 3, 1).reshape(O, kh * kw * C).to(cdt)  # K index = tap*C + c
            gcolT = torch.mm(go2.t(), w_ok)  # (B*hw, kh*kw*C): channels-last column gradient
            if dg == 1:  # gather form: no floating-point atomics (3.4x faster at pyramid level 0)
                gi = deformable_col2im_gather_nhwc(gcolT, offset, (B, H, W, C), (kh, kw), padding, stride, dilation,
                                                   slot=getattr(ctx, "slot", None))
            else:
                gi = deformable_col2im_nhwc(gcolT, offset, (B, H, W, C), (kh, kw), padding, stride, dilation, dg)
            grad_input = nhwc_to_nchw(gi)
        if ctx.needs_input_grad[2]:
            # gw (O, C*kh*kw) = go2 @ col^T has only 36 output tiles of 128 x 128 for K = B*hw up to 65 536: split K into
            # J slices as a strided-batched product (views, no copies) and add the J partial results
            n = B * hw
            J = 16 if n % 16 == 0 and n >= 4096 else 1
            k = n // J
            if ctx.col_is_T:   # columns (positions, tap*C + c) from the implicit-GEMM forward
                if col is None:
                    raise RuntimeError("AlignConv: the weight gradient needs the columns of a forward run with grad enabled")
                parts = torch.bmm(go2.view(O, J, k).permute(1, 0, 2), col.view(J, k, kh * kw * C))
                gw = parts.sum(0, dtype=torch.float32) if J > 1 else parts[0].float()
                grad_weight = gw.view(O, kh, kw, C).permute(0, 3, 1, 2).contiguous()
            else:
                parts = torch.bmm(go2.view(O, J, k).permute(1, 0, 2), col.view(C * kh * kw, J, k).permute(1, 2, 0))
                grad_weight = (parts.sum(0, dtype=torch.float32) if J > 1 else parts[0].float()).view_as(weight)
        return grad_input, None, grad_weight, None, None, None, None, None


def _dcn_backward_channels_last(ctx, grad_output):
    """Backward of the channels_last fp32 implicit-GEMM forward: the same three products as the NCHW form on
    (positions, channels) views of the channels_last gradient -- no transposed copies -- and a channels_last result."""
    offset, weight, colT = ctx.saved_tensors
    stride, padding, dilation, dg = ctx.cfg
    B, C, H, W = ctx.in_shape
    O, _, kh, kw = weight.shape
    go = grad_output.permute(0, 2, 3, 1)
    if not go.is_contiguous():
        go = go.contiguous()
    go = go.reshape(-1, O).to(weight.dtype)                       # (P, O)
    grad_input = grad_weight = None
    if ctx.needs_input_grad[0]:
        w_ok = weight.permute(0, 2, 3, 1).reshape(O, kh * kw * C)
        gcolT = torch.mm(go, w_ok)                                 # (P, kh*kw*C)
        if dg == 1:
            gi = deformable_col2im_gather_nhwc(gcolT, offset, (B, H, W, C), (kh, kw), padding, stride, dilation,
                                                   slot=getattr(ctx, "slot", None))
        else:
            gi = deformable_col2im_nhwc(gcolT, offset, (B, H, W, C), (kh, kw), padding, stride, dilation, dg)
        grad_input = gi.permute(0, 3, 1, 2)                        # channels_last storage, NCHW shape
    if ctx.needs_input_grad[2]:
        if colT is None:
            raise RuntimeError("AlignConv: the weight gradient needs the columns of a forward run with grad enabled")
        P = go.shape[0]
        J = 16 if P % 16 == 0 and P >= 4096 else 1
        parts = torch.bmm(go.view(J, P // J, O).transpose(1, 2), colT.view(J, P // J, kh * kw * C))
        gw = parts.sum(0, dtype=torch.float32) if J > 1 else parts[0].float()
        grad_weight = gw.view(O, kh, kw, C).permute(0, 3, 1, 2).contiguous()
    return grad_input, None, grad_weight, None, None, None, None, None


DeformConvFunctionNHWC._backward_channels_last = staticmethod(_dcn_backward_channels_last)


# AlignConv on the matrix cores as an implicit GEMM (csrc/alignconv_mfma.hip) wherever _mfma_geom covers the call; the
# im2col + rocBLAS form serves the other geometries (tests set this False to compare the two)
_MFMA_ALIGNCONV = True


def _mfma_geom(input, weight, stride, padding, dilation, deformable_groups):
    """The rsdet_dcn_geom of the call if the implicit-GEMM kernel covers it, else None."""
    B, C, H, W = input.shape
    O, _, kh, kw = weight.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_groups)
    return g if _lib.load().rsdet_alignconv_mfma_supported(g, O) else None


def _mfma_geom_f32(input, weight, stride, padding, dilation, deformable_groups, min_tiles=384):
    """Exact-fp32 implicit GEMM: 1/16 of the bf16 matrix rate, so it only pays where the launch fills the chip (one
    workgroup per 128 positions runs 72 K steps of ~5 us; pyramid level 0 of a 4-tile batch has 512 of them)."""
    B, C, H, W = input.shape
    O, _, kh, kw = weight.shape
    if kh != 3 or kw != 3:
        return None
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ho, Wo = H + 2 * ph - 2, W + 2 * pw - 2
    if B * ((Ho + 7) // 8) * ((Wo + 15) // 16) * ((O + 255) // 256) < min_tiles:
        return None
    g = _geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, deformable_groups)
    return g if _lib.load().rsdet_alignconv_mfma_f32_supported(g, O) else None


class AlignConvMFMAFunction(torch.autograd.Function):
    """bf16-autocast form of DeformConvFunction.forward / backward (dcn_v1.py:412-557) for the AlignConv geometry:
    forward is ONE launch (samples interpolated in fp32 from the bf16 channels-last activations, rounded to bf16 in
    LDS, products on v_mfma_f32_32x32x16_bf16; the sampled columns are written out once, channels-last, for the weight
    gradient); backward = two bf16 GEMMs on views + the gather-form col2im.  No cast of the input: a channels_last
    bf16 tensor is consumed as it is and the result is channels_last bf16."""

    @staticmethod
    def forward(ctx, input, offset, weight, geom, padding):
        lib = _lib.load()
        B, C, H, W = input.shape
        O = weight.shape[0]
        Ho, Wo = H + 2 * padding[0] - 2, W + 2 * padding[1] - 2
        x = input.permute(0, 2, 3, 1)                      # (B,H,W,C): a view of a channels_last tensor
        if x.dtype != torch.bfloat16 or not x.is_contiguous():
            x = x.to(torch.bfloat16).contiguous()
        off = offset.float().contiguous()
        # k = tap*C + c; the levels of a step share the weight (an fp32 parameter: DeformConv is no nn.Conv2d, the Runner
        # leaves it fp32): cast once per forward pass, not once per level
        key = (id(weight), weight._version)
        hit = _PLAN.operands.get(key) if _PLAN is not None else None
        if hit is not None and hit[0] is weight:
            w_flat = hit[1]
        else:
            w_flat = weight.detach().permute(0, 2, 3, 1).reshape(O, 9 * C).to(torch.bfloat16)
            if _PLAN is not None:
                _PLAN.operands[key] = (weight, w_flat)
        ctx.w_cl = weight.dim() == 4 and not weight.is_contiguous() and weight.is_contiguous(memory_format=torch.channels_last)
        need_w = ctx.needs_input_grad[2]
        # channels_last storage == (B,Ho,Wo,O); returned as is (callers apply ReLU in place: not a view)
        out = torch.empty((B, O, Ho, Wo), dtype=torch.bfloat16, device=input.device,
                          memory_format=torch.channels_last)
        colT = torch.empty((B * Ho * Wo, 9 * C), dtype=torch.bfloat16, device=input.device) if need_w else None
        _lib.check(lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x), _lib.ptr(off), _lib.ptr(w_flat), geom, O, 1,
                                                     _lib.ptr(out), _lib.ptr(colT), _lib.stream_ptr()),
                   "rsdet_alignconv_fwd_mfma_bf16")
        ctx.save_for_backward(off, w_flat, colT)
        ctx.shape = (B, C, H, W, O, Ho, Wo)
        ctx.padding = padding
        ctx.in_dtype = input.dtype
        ctx.slot = _plan_slot(ctx, off, C, H, W, (3, 3), tuple(padding), (1, 1), (1, 1), B, 1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        off, w_flat, colT = ctx.saved_tensors
        B, C, H, W, O, Ho, Wo = ctx.shape
        go = grad_output.permute(0, 2, 3, 1)               # (B,Ho,Wo,O): a view of a channels_last gradient
        if go.dtype != torch.bfloat16 or not go.is_contiguous():
            go = go.to(torch.bfloat16).contiguous()
        go = go.reshape(B * Ho * Wo, O)
        grad_input = grad_weight = None
        if ctx.needs_input_grad[0]:
            gcolT = torch.mm(go, w_flat)                   # (P, 9*C): channels-last column gradient, bf16
            gi = deformable_col2im_gather_nhwc(gcolT, off, (B, H, W, C), (3, 3), ctx.padding, (1, 1), (1, 1),
                                               slot=ctx.slot, out_dtype=ctx.in_dtype)
            grad_input = gi.permute(0, 3, 1, 2).to(ctx.in_dtype)       # (no-op when the gather stored bf16 itself)
        if ctx.needs_input_grad[2]:
            if colT is None:
                raise RuntimeError("AlignConv: the weight gradient needs the columns of a forward run with grad enabled")
            P = go.shape[0]
            J = 16 if P % 16 == 0 and P >= 4096 else 1     # split K = P: 36 output tiles only otherwise
            parts = torch.bmm(go.view(J, P // J, O).transpose(1, 2), colT.view(J, P // J, 9 * C))
            gw = parts.sum(0, dtype=torch.float32) if J > 1 else parts[0].float()
            grad_weight = gw.view(O, 3, 3, C).permute(0, 3, 1, 2)      # = the channels_last storage of an (O,C,3,3) tensor
            if not ctx.w_cl:                                            # a channels_last weight takes it as it is
                grad_weight = grad_weight.contiguous()
        return grad_input, None, grad_weight, None, None


class DeformConvFunction(torch.autograd.Function):
    """dcn_v1.py:559-650 (reference column layout; used for groups > 1 or when the offset needs a gradient)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        if not input.is_cuda:
            raise NotImplementedError  # dcn_v1.py:588-589
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        B, C, H, W = input.shape
        O, _, kh, kw = weight.shape
        Ho, Wo = _out_hw(H, W, kh, kw, *ctx.padding, *ctx.stride, *ctx.dilation)
        if not (O > 0 and Ho > 0 and Wo > 0):
            raise ValueError("convolution input is too small (output would be {})".format(
                'x'.join(map(str, (B, O, Ho, Wo)))))
        assert offset.size(0) == B, "invalid batch size of offset"
        col = deformable_im2col(input, offset, (kh, kw), ctx.padding, ctx.stride, ctx.dilation, deformable_groups)
        # (g, O/g, C/g*kh*kw) @ (g, C/g*kh*kw, B*Ho*Wo)
        wg = weight.reshape(groups, O // groups, -1)
        out = torch.bmm(wg, col.view(groups, -1, col.shape[1]))
        out = out.view(O, B, Ho, Wo).transpose(0, 1).contiguous()
        ctx.save_for_backward(input, offset, weight, col)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        input, offset, weight, col = ctx.saved_tensors
        B, C, H, W = input.shape
        O, _, kh, kw = weight.shape
        g = ctx.groups
        go = grad_output.transpose(0, 1).reshape(g, O // g, -1)  # (g, O/g, B*Ho*Wo)
        grad_input = grad_offset = grad_weight = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wg = weight.reshape(g, O // g, -1)
            gcol = torch.bmm(wg.transpose(1, 2), go).view(C * kh * kw, -1)  # dcn_v1.py:484
            if ctx.needs_input_grad[1]:
                grad_offset = deformable_col2im_coord(gcol, input, offset, (kh, kw), ctx.padding, ctx.stride,
                                                      ctx.dilation, ctx.deformable_groups)
            if ctx.needs_input_grad[0]:
                grad_input = deformable_col2im(gcol, offset, input.shape, (kh, kw), ctx.padding, ctx.stride,
                                               ctx.dilation, ctx.deformable_groups)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.bmm(go, col.view(g, -1, col.shape[1]).transpose(1, 2)).view_as(weight)  # :547
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    """dcn_v1.py:650 ``deform_conv = DeformConvFunction.apply`` (same positional signature)."""
    if groups == 1 and not offset.requires_grad and input is not None and input.dim() == 4 and input.is_cuda:
        # bf16 columns + bf16 products under autocast: 26.8 -> 25.8 ms per channels_last S2ANet step (24.1 together with
        # the fused bias + ReLU of the towers); in round 1 the step was host-bound and the same switch bought nothing
        lowp = (_LOWP_ALIGNCONV and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if lowp and _MFMA_ALIGNCONV:
            geom = _mfma_geom(input, weight, stride, padding, dilation, deformable_groups)
            if geom is not None:
                return AlignConvMFMAFunction.apply(input, offset, weight, geom, _pair(padding))
        return DeformConvFunctionNHWC.apply(input, offset, weight, stride, padding, dilation, deformable_groups, lowp)
    return DeformConvFunction.apply(input, offset, weight, stride, padding, dilation, groups, deformable_groups,
                                    im2col_step)


class DeformConv(nn.Module):
    """dcn_v1.py:652-695."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, 'in_channels {} cannot be divisible by groups {}'.format(in_channels, groups)
        assert out_channels % groups == 0, 'out_channels {} cannot be divisible by groups {}'.format(out_channels, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.zeros((out_channels, in_channels // groups, *self.kernel_size)))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        nn.init.uniform_(self.weight, -stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)
