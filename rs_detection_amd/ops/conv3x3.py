"""3x3 / stride-1 / padding-1 convolutions with as many outputs as inputs: backward-data through the FORWARD solver.

The gradient of such a convolution with respect to its input is itself a 3x3 / stride-1 / padding-1 convolution of the
output gradient -- with the weights flipped in both spatial directions and their channel axes exchanged -- and of the
SAME problem size, so MIOpen answers it with the same tuned forward solver.  Measured on the shapes of the 4 x 1024^2
step (profiles/r03_conv3x3_bwd.txt): MIOpen's backward-data solvers take 1.35-1.65 x the time of its forward solver in
bf16 (the head canvas: 245 vs 147 us) and 1.05-1.2 x in fp32 (level 0: 702 vs 591 us); the flipped weights cost one
2.4 MB copy per call.  The weight gradient stays with MIOpen's own solver.

Same arithmetic as the reference's convolution backward (each input-gradient element is the same sum of products,
accumulated in fp32); which library kernel forms it is not part of the semantics
(/root/reference/python/jdet/models/roi_heads/s2anet_head.py:207-252 tower convolutions, necks/fpn.py, the Bottleneck's
conv2 in models/backbones/resnet.py:101-126)."""
import os

import torch
import torch.nn.functional as F

_ON = os.environ.get("RSDET_CONV3X3_BWD_AS_FWD", "1") == "1"   # A/B switch
# fp32: measured neutral on the step (50.5 vs 50.5 ms: MIOpen's fp32 backward-data solver is as fast as its forward one
# there), so only bf16 takes this route by default
_F32 = os.environ.get("RSDET_CONV3X3_BWD_AS_FWD_F32", "0") == "1"


def _flipped(w):
    """(O, C, 3, 3) -> (C, O, 3, 3) with both spatial axes reversed, channels_last like the tensors around it: one launch
    (csrc/layout.hip: weight_flip_transpose_kernel) instead of torch's flip + strided copy (12 -> 3 us)."""
    from .. import _lib
    O, C, kh, kw = w.shape
    wc = w.contiguous(memory_format=torch.channels_last)            # storage (O, kh*kw, C)
    out = torch.empty((C, O, kh, kw), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
    rc = _lib.load().rsdet_weight_flip_transpose(_lib.ptr(wc), _lib.ptr(out), O, C, kh * kw, w.element_size(),
                                                 _lib.stream_ptr())
    _lib.check(rc, "rsdet_weight_flip_transpose")
    return out


class _Conv3x3Same(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.bfloat16)
    def forward(ctx, x, w, bias):
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        y = F.conv2d(x, w, None if bias is None else bias.to(x.dtype), 1, 1)
        ctx.save_for_backward(x, w)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = F.conv2d(gy, _flipped(w), None, 1, 1)
        if ctx.needs_input_grad[1]:
            gw = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     (False, True, False))[1]
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3), dtype=torch.float32).to(ctx.bias_dtype)
        return gx, gw, gb


def conv3x3_applies(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1):
    return (_ON and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and weight.shape[0] == weight.shape[1]
            and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and groups == 1
            and x.is_cuda and x.dim() == 4 and x.shape[1] == weight.shape[1]
            and (x.dtype == torch.bfloat16 or (_F32 and x.dtype == torch.float32 and not torch.is_autocast_enabled()))
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and torch.is_grad_enabled() and x.requires_grad)


def conv3x3_same(x, weight, bias=None):
    """``F.conv2d(x, weight, bias, 1, 1)`` for a 3x3 weight; backward-data through the forward solver when the layer is
    square (C_out == C_in), the tensor channels_last on the GPU and a gradient of x is wanted -- plain conv2d otherwise."""
    if conv3x3_applies(x, weight):
        return _Conv3x3Same.apply(x, weight, bias)
    return F.conv2d(x, weight, bias, 1, 1)


def fast_conv(conv, x):
    """``conv(x)`` for an ``nn.Conv2d``: the GEMM split of ops/conv1x1.py for 1x1 / stride-1 layers, the rule above for
    square 3x3 / stride-1 / padding-1 layers, the module itself otherwise."""
    from .conv1x1 import conv1x1, conv1x1_applies
    if type(conv) is torch.nn.Conv2d and conv.padding_mode == 'zeros':
        if conv1x1_applies(conv, x):
            return conv1x1(conv, x)
        if conv3x3_applies(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
            return _Conv3x3Same.apply(x, conv.weight, conv.bias)
    return conv(x)
