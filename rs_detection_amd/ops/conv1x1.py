"""1x1 / stride-1 convolutions of a channels_last step as what they are: GEMMs on (pixels, channels) views.

A channels_last (B, C, H, W) tensor IS the row-major (B*H*W, C) matrix, so ``conv1x1(x, w) = x2d @ w2d^T`` and its
backward-data ``gy2d @ w2d`` need no copies.  MIOpen runs these through its implicit-GEMM convolution kernels; measured on
the ResNet-50 / FPN shapes of the 4 x 1024^2 step (profiles/r03_conv1x1.txt, bf16): backward-data is 1.3-1.9 x faster as a
library GEMM everywhere, the forward is faster for maps of <= 128^2 (the FPN laterals: 33-84 us -> < 19 us) and slower at
256^2, and the weight gradient (K = B*H*W, a split-K reduction) is 5-10 x SLOWER as a GEMM.  Hence the split:

    forward        GEMM for B*H*W <= 65 536 pixels, MIOpen above
    backward-data  GEMM (torch.mm -> hipBLASLt / rocBLAS)
    backward-w     split-K batched GEMM on views since round 4 (_wrw_split_k); MIOpen before

The arithmetic is the convolution's own (products accumulated in fp32, one rounding of the result); which library
computes it is not part of the reference's semantics (the reference calls cuDNN here:
/root/reference/python/jdet/models/backbones/resnet.py:101-126, necks/fpn.py lateral convolutions)."""
import os

import torch
import torch.nn.functional as F

from ._amp import light_custom_bwd, light_custom_fwd

_ON = True
_FWD_MAX_PIXELS = 65536


_WRW_GEMM = True    # the weight gradient as a split-K batched GEMM (False: MIOpen's kernel)


def _wrw_split_k(gy2, x, w, rowscale=None, rowdot=False):
    """Weight gradient gw[o, c] = sum_p gy2[p, o] x2[p, c] as a SPLIT-K batched GEMM on views: the pixel axis cut into S
    slices, one (O, P/S) x (P/S, C) product per slice (torch.bmm on views, no copies), the S partial results summed.  A
    plain GEMM has K = all pixels and only (O/256)(C/256) tiles to spread over 256 CUs -- 5-10 x slower than MIOpen;
    with S ~ P / 1024 (bf16) or P / 2048 (fp32) slices it is 25-35 % (bf16: 36 -> 25 us per call on the trunk shapes)
    and 5-12 % (fp32) faster than MIOpen's kernel + zero-fill (+ cast) launches (scratch/wrw1x1_nhwc.py, round 4).
    The S partial products come back in fp32 (``out_dtype``: the GEMM's own accumulator, not rounded to bf16) and are
    summed in fp32: ONE rounding of the result, as in MIOpen's kernel.
    ``rowscale = (running_var, gamma, eps)``: row o of the result times gamma[o] / sqrt(running_var[o] + eps) before the
    rounding -- gy2 is then the gradient of the OUTPUT of an eval-mode BatchNorm behind the convolution (ops/bottleneck.py).
    ``rowdot`` (with rowscale): returns ``(gw, d)`` with d[o] = sum_c w[o, c] U[o, c], U the unscaled fp32 gradient =
    sum_p gy2[p, o] conv(x)[p, o] -- the term that BatchNorm's scale gradient is formed from without the convolution's
    output ever having been stored (csrc/bn_act.hip: rsdet_bn_affine_grads_finish_multi_f32)."""
    P, O = gy2.shape
    C = x.shape[1]
    S = max(1, min(64, P // (1024 if x.dtype == torch.bfloat16 else 2048)))
    while S > 1 and P % S:
        S //= 2
    x2 = x.permute(0, 2, 3, 1).reshape(P, C)
    assert rowscale is not None or not rowdot

    def cl(t):
        return t.view(O, C, 1, 1).contiguous(memory_format=torch.channels_last)
    if rowscale is not None and not (x.dtype == torch.bfloat16 and C % 4 == 0 and w.dtype == torch.bfloat16
                                     and w.is_contiguous(memory_format=torch.channels_last)):
        var, gamma, eps = rowscale
        sc = torch.rsqrt(var + eps) * (1.0 if gamma is None else gamma)
        u = torch.mm(gy2.t(), x2).float()
        gw = cl((u * sc[:, None]).to(x.dtype))
        return (gw, (u * w.reshape(O, C).float()).sum(1)) if rowdot else gw
    if S == 1 and rowscale is None:
        return cl(torch.mm(gy2.t(), x2))
    a, b = gy2.view(S, P // S, O).transpose(1, 2), x2.view(S, P // S, C)
    part = torch.bmm(a, b, out_dtype=torch.float32) if x.dtype == torch.bfloat16 else torch.bmm(a, b)
    if part.dtype == torch.float32 and (O * C) % 4 == 0:
        # fold of the S fp32 partial products as one pass of our own (fp32 sums, one rounding, written in the weight's
        # channels_last storage = the (O, C) matrix) instead of a reduction launch + a cast launch
        from .. import _lib
        gw = torch.empty((O, C, 1, 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        if rowscale is not None:
            var, gamma, eps = rowscale
            d = torch.empty((O,), dtype=torch.float32, device=x.device) if rowdot else None
            rc = _lib.load().rsdet_sum_slabs_rowscale_f32(_lib.ptr(part), S, O * C, C, _lib.ptr(var), _lib.ptr(gamma),
                                                          float(eps), _lib.ptr(w) if rowdot else None, _lib.ptr(d),
                                                          _lib.ptr(gw), int(x.dtype == torch.bfloat16), _lib.stream_ptr())
            _lib.check(rc, "rsdet_sum_slabs_rowscale_f32")
            return (gw, d) if rowdot else gw
        rc = _lib.load().rsdet_sum_slabs_f32(_lib.ptr(part), S, O * C, _lib.ptr(gw), int(x.dtype == torch.bfloat16),
                                             _lib.stream_ptr())
        _lib.check(rc, "rsdet_sum_slabs_f32")
        return gw
    return cl(part.sum(0).to(x.dtype))


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    @light_custom_fwd(torch.bfloat16)
    def forward(ctx, x, w, bias):
        B, C, H, W = x.shape
        O = w.shape[0]
        if not (x.dtype == w.dtype):
            w = w.to(x.dtype)
        x = x.contiguous(memory_format=torch.channels_last)
        P = B * H * W
        if P <= _FWD_MAX_PIXELS:
            x2, w2 = x.permute(0, 2, 3, 1).reshape(P, C), w.reshape(O, C)
            y2 = torch.mm(x2, w2.t()) if bias is None else torch.addmm(bias.to(x.dtype), x2, w2.t())
            y = y2.view(B, H, W, O).permute(0, 3, 1, 2)
        else:
            y = F.conv2d(x, w, None if bias is None else bias.to(x.dtype))
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    @light_custom_bwd
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, C, H, W = x.shape
        O = w.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gy2 = gy.permute(0, 2, 3, 1).reshape(-1, O)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(gy2, w.reshape(O, C)).view(B, H, W, C).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = _wrw_split_k(gy2, x, w) if _WRW_GEMM else None
            if gw is None:
                gw = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        return gx, gw, gb


def conv1x1_applies(conv, x):
    return (_ON and type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last))


def conv1x1(conv, x):
    """``conv(x)`` for an ``nn.Conv2d``; the GEMM split above when it is a 1x1 / stride-1 convolution of a channels_last
    CUDA tensor, the module itself otherwise."""
    if conv1x1_applies(conv, x):
        return _Conv1x1.apply(x, conv.weight, conv.bias)
    return conv(x)


# --------------------------------------------------------------------------------------------------------------------
# NCHW fp32 maps (the VAN backbone of Oriented R-CNN): the weight gradient of a 1x1 convolution on LARGE maps as a
# split-K batched GEMM on strided views.  gw = sum_{n, s} gy[n][:, K-slice s] @ x[n][:, K-slice s]^T: both operands are
# K-contiguous as they lie (no layout change), where MIOpen's best solver for this layout is an NHWC kernel between two
# transposes and a zero-fill.  Device time per call (profiles/README.md, round 4; 2 tiles): 64 -> 512 at 256^2 206 ->
# 94 us, 128 -> 1024 at 128^2 146 -> 87 us, 64 -> 64 46 -> 32 us; from 64^2 maps down MIOpen is as fast or faster and
# keeps the call.  Forward and backward-data stay with MIOpen.
_NCHW_WRW = True
_NCHW_WRW_MIN_PIXELS = 128 * 128
_NCHW_WRW_SPLITS = 16


# forward / backward-data of the NCHW 1x1 convolutions as batched GEMMs on views instead of MIOpen's 1x1 solver (which
# launches the same kind of kernel behind a ~23 us host path).  Measured on the Oriented R-CNN step, same box, twice
# (profiles/r05_orcnn_mm.txt): 65.5 / 65.9 ms through MIOpen, 66.8 / 66.5 ms as bmm on a stride-0 batch view of the
# weight (torch.matmul(2-D, 3-D) COPIES the map: 92 ms) -- host time equal, the library's kernel choice slightly
# better.  Off; the switch is what the equivalence test flips.
_NCHW_MM = False


class _Conv1x1NCHW(torch.autograd.Function):
    """Bias-free 1x1 / stride-1 convolution of an NCHW fp32 map (the VAN block's five per block,
    /root/reference/python/jdet/models/backbones/van.py:46-122): MIOpen forward and backward-data (or, with _NCHW_MM,
    batched GEMMs on views: an NCHW map IS N row-major (C, H W) matrices), weight gradient as the split-K batched GEMM
    below on large maps, MIOpen's kernel on small ones."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if not _NCHW_MM:
            return F.conv2d(x, w, None)
        N, C, H, W = x.shape
        O = w.shape[0]
        # (bmm on a stride-0 batch view of the weight: torch.matmul(2-D, 3-D) folds the batch into the GEMM's N by COPYING
        #  the map -- measured 92 ms per step against 67)
        return torch.bmm(w.view(1, O, C).expand(N, O, C), x.view(N, C, H * W)).view(N, O, H, W)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        N, C, H, W = x.shape
        O, HW = w.shape[0], H * W
        if ctx.needs_input_grad[0]:
            if _NCHW_MM:
                gx = torch.bmm(w.view(O, C).t().unsqueeze(0).expand(N, C, O), gy.view(N, O, HW)).view(N, C, H, W)
            else:
                gx = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1] and not (HW >= _NCHW_WRW_MIN_PIXELS and HW % (_NCHW_WRW_SPLITS * 64) == 0):
            gw = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                     (False, True, False))[1]
        elif ctx.needs_input_grad[1]:
            S = _NCHW_WRW_SPLITS
            k = HW // S
            part = torch.empty((N * S, O, C), dtype=x.dtype, device=x.device)
            for n in range(N):
                a = gy[n].view(O, S, k).permute(1, 0, 2)            # (S, O, k): strides (k, HW, 1)
                b = x[n].view(C, S, k).permute(1, 2, 0)             # (S, k, C): strides (k, 1, HW)
                torch.bmm(a, b, out=part[n * S:(n + 1) * S])
            gw = part.sum(0).view(O, C, 1, 1)
        return gx, gw


def conv1x1_nchw(x, weight):
    """``F.conv2d(x, weight)`` for a bias-free 1x1 convolution; on NCHW fp32 CUDA maps outside autocast forward and
    backward-data are batched GEMMs on views, and on large maps the weight gradient is the split-K batched GEMM above."""
    if (_NCHW_WRW and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.is_contiguous() and not torch.is_autocast_enabled() and torch.is_grad_enabled() and weight.requires_grad
            and tuple(weight.shape[2:]) == (1, 1) and weight.is_contiguous()
            and (_NCHW_MM or (x.shape[2] * x.shape[3] >= _NCHW_WRW_MIN_PIXELS
                              and (x.shape[2] * x.shape[3]) % (_NCHW_WRW_SPLITS * 64) == 0))):
        return _Conv1x1NCHW.apply(x, weight)
    return F.conv2d(x, weight, None)
