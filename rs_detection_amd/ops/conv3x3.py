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

from . import weight_prep as wprep
from ._amp import light_custom_bwd, light_custom_fwd

_ON = True
# fp32: measured neutral on the step (50.5 vs 50.5 ms: MIOpen's fp32 backward-data solver is as fast as its forward one
# there), so only bf16 takes this route
_F32 = False


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
    @light_custom_fwd(torch.bfloat16)
    def forward(ctx, x, w, bias):
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        y = F.conv2d(x, w, None if bias is None else bias.to(x.dtype), 1, 1)
        ctx.save_for_backward(x, w)
        # a bf16 Parameter: its flipped form comes from the once-per-optimizer-step launch of ops/weight_prep.py
        ctx.prep = wprep.entry(w, flip=True) if wprep.applies(w) else None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    @light_custom_bwd
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = F.conv2d(gy, ctx.prep.tensor() if ctx.prep is not None else _flipped(w), None, 1, 1)
        if ctx.needs_input_grad[1]:
            # bf16 layers whose channel counts fill the kernel's 256-wide tiles: our split-K weight gradient (two
            # launches, no zero-fill / cast launches around it)
            if (_WRW_TRUNK and x.dtype == torch.bfloat16 and w.shape[0] % _WRW_TRUNK_O == 0
                    and w.shape[1] % 64 == 0):
                gw = _mfma_wrw(gy, x, w.dtype)
            if gw is None:
                gw = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3), dtype=torch.float32).to(ctx.bias_dtype)
        return gx, gw, gb


# --------------------------------------------------------------------------------------------------------------------
# The head canvas in bf16: conv + bias + ReLU + gap mask as ONE launch of our own implicit-GEMM kernel
# (csrc/conv3x3_mfma.hip), backward-data through the same kernel on the flipped weights.
_MFMA = True
# the same weight-gradient kernel for the other square bf16 3x3 layers (ResNet conv2 of layers 2-4, the FPN output
# convolutions): bf16 step 16.03 -> 15.88 ms with output-channel multiples of 128 (half-filled 256-wide tiles still beat
# MIOpen's kernel + zero-fill + cast launches there), 15.90 with multiples of 256 only (profiles/README.md, round 4)
_WRW_TRUNK = True
_WRW_TRUNK_O = 128                                              # output-channel multiple it takes
_MFMA_TM = 224                                                  # positions of one row a workgroup covers (C3_TM)


def _mfma_conv(x, w_cl, bias, live, relu):
    """x (B,C,H,W) bf16 channels_last, w_cl (O,C,3,3) bf16 channels_last -> (B,O,H,W) bf16 channels_last."""
    from .. import _lib
    B, C, H, W = x.shape
    O = w_cl.shape[0]
    y = torch.empty((B, O, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    rc = _lib.load().rsdet_conv3x3_fwd_mfma_bf16(_lib.ptr(x), _lib.ptr(w_cl), _lib.ptr(bias), _lib.ptr(live), B, H, W, C,
                                                 O, int(relu), _lib.ptr(y), _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv3x3_fwd_mfma_bf16")
    return y


_MFMA_WRW = True


def _mfma_wrw(g, x, out_dtype, rowscale=None, weight=None):
    """Weight gradient (O, C, 3, 3) channels_last of the 3x3 convolution from g (B,O,H,W) and x (B,C,H,W), both bf16
    channels_last (csrc/conv3x3_wrw_mfma.hip: split-K implicit GEMM + fold, two launches); None when the kernel does not
    take the shape or the weight's dtype.
    ``rowscale = (running_var, gamma, eps)`` with ``weight`` (the convolution's own bf16 channels_last weight): g is the
    gradient of the OUTPUT of an eval-mode BatchNorm behind the convolution -- row o of the result is scaled by gamma[o] /
    sqrt(var[o] + eps) and ``(gw, rowdot)`` is returned, rowdot[o] = sum weight[o] * (the unscaled fp32 row) (the term
    that BatchNorm's scale gradient is formed from, ops/bottleneck.py)."""
    from .. import _lib
    if out_dtype not in (torch.bfloat16, torch.float32):
        return None
    lib = _lib.load()
    B, C, H, W = x.shape
    O = g.shape[1]
    if not lib.rsdet_conv3x3_wrw_mfma_supported(B, H, W, C, O):
        return None
    gw = torch.empty((O, C, 3, 3), dtype=out_dtype, device=x.device, memory_format=torch.channels_last)
    nb = lib.rsdet_conv3x3_wrw_mfma_ws_size(B, H, W, C, O)
    ws = torch.empty((nb,), dtype=torch.uint8, device=x.device)
    if rowscale is not None:
        var, gamma, eps = rowscale
        d = torch.empty((O,), dtype=torch.float32, device=x.device)
        rc = lib.rsdet_conv3x3_wrw_mfma_rowscale_bf16(_lib.ptr(g), _lib.ptr(x), B, H, W, C, O, _lib.ptr(var),
                                                      _lib.ptr(gamma), float(eps), _lib.ptr(weight), _lib.ptr(d),
                                                      _lib.ptr(gw), int(out_dtype == torch.bfloat16), _lib.ptr(ws), nb,
                                                      _lib.stream_ptr())
        _lib.check(rc, "rsdet_conv3x3_wrw_mfma_rowscale_bf16")
        return gw, d
    rc = lib.rsdet_conv3x3_wrw_mfma_bf16(_lib.ptr(g), _lib.ptr(x), B, H, W, C, O, _lib.ptr(gw),
                                         int(out_dtype == torch.bfloat16), _lib.ptr(ws), nb, _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv3x3_wrw_mfma_bf16")
    return gw


def _lib_supported(B, H, W, C, O):
    from .. import _lib
    return bool(_lib.load().rsdet_conv3x3_mfma_supported(B, H, W, C, O))


def _bias_relu_backward(gy, y, need_b):
    """gy * (y > 0) and its per-channel fp32 sum in one launch (csrc/bn_act.hip with mean 0 / variance 1)."""
    from .. import _lib
    from .bn_act import _UNIT
    lib = _lib.load()
    N, C, H, W = gy.shape
    if not lib.rsdet_bn_act_nhwc_supported(C):        # a channel count the fused pass does not tile: two torch passes
        gx = gy * (y > 0)
        return gx, (gx.sum((0, 2, 3), dtype=torch.float32) if need_b else None)
    gx = torch.empty_like(gy)
    gb = torch.empty((C,), dtype=torch.float32, device=gy.device) if need_b else None
    ws_bytes = lib.rsdet_bn_act_backward_nhwc_ws_size(N, C, H * W) if need_b else 0
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gy.device) if ws_bytes else None
    key = (gy.device, C)
    if key not in _UNIT:
        _UNIT[key] = (torch.zeros(C, device=gy.device), torch.ones(C, device=gy.device))
    mean, var = _UNIT[key]
    rc = lib.rsdet_bn_act_backward_nhwc_bf16(_lib.ptr(gy), _lib.ptr(y), None, _lib.ptr(mean), _lib.ptr(var), None, 0.0, N,
                                             C, H * W, 1, _lib.ptr(gx), None, None, _lib.ptr(gb), _lib.ptr(ws), ws_bytes,
                                             _lib.stream_ptr())
    _lib.check(rc, "rsdet_bn_act_backward_nhwc_bf16")
    return gx, gb


class _Conv3x3BiasReLU(torch.autograd.Function):
    """relu(conv3x3(x, w) + bias) with the canvas gap pixels written as zeros -- one launch forward.  Backward: the ReLU
    gate + bias gradient (one launch; the gate y > 0 also excludes the gap pixels), backward-data as the same kernel
    on the flipped weights, the weight gradient from MIOpen.  Accumulation in fp32, bf16 products and one rounding of
    the result, like MIOpen's bf16 solvers (identical values on the shapes of tests/test_gpu_ops.py)."""

    @staticmethod
    def forward(ctx, x, w, bias, live):
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        xb = xb.contiguous(memory_format=torch.channels_last)
        wb = (w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)).contiguous(memory_format=torch.channels_last)
        bf = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        y = _mfma_conv(xb, wb, bf, live, True)
        ctx.save_for_backward(xb, wb, y)
        ctx.prep = wprep.entry(w, flip=True) if (wb is w and wprep.applies(w)) else None
        ctx.in_dtype, ctx.w_dtype = x.dtype, w.dtype
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb, y = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        need_b = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        g, gb = _bias_relu_backward(gy, y, need_b)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wf = ctx.prep.tensor() if ctx.prep is not None else _flipped(wb)
            B, O, H, W = g.shape
            if _lib_supported(B, H, W, O, wf.shape[0]):
                gx = _mfma_conv(g, wf, None, None, False)
            else:                     # the transposed problem's channel counts do not tile (C_out % 64): MIOpen's solver
                gx = F.conv2d(g, wf, None, 1, 1)
            if gx.dtype != ctx.in_dtype:
                gx = gx.to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            gw = _mfma_wrw(g, xb, ctx.w_dtype) if _MFMA_WRW else None
            if gw is None:           # MIOpen's solver (+ its zero-fill and cast launches)
                gw = torch.ops.aten.convolution_backward(g, xb, wb, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
                if gw.dtype != ctx.w_dtype:
                    gw = gw.to(ctx.w_dtype)
        if gb is not None and gb.dtype != ctx.bias_dtype:
            gb = gb.to(ctx.bias_dtype)
        return gx, gw, gb, None


def conv3x3_mfma_applies(x, conv):
    """Our kernel takes the layer: bf16 (or bf16 autocast) channels_last CUDA map, 3x3 / stride 1 / padding 1 / no groups,
    channel counts its tiles divide, and a row length its 224-position tile wastes at most 15 % of (the 196-wide head
    canvas of 1024^2 tiles: 12.5 %; a 128-wide level alone would idle 43 % of the MFMAs -- MIOpen keeps those)."""
    if not (_MFMA and type(conv) is torch.nn.Conv2d and conv.padding_mode == 'zeros' and x.is_cuda and x.dim() == 4):
        return False
    w = conv.weight
    if not (tuple(w.shape[2:]) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and x.shape[1] == w.shape[1]):
        return False
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and x.dtype == torch.float32
                                          and torch.get_autocast_dtype('cuda') == torch.bfloat16)):
        return False
    if x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    B, C, H, W = x.shape
    O = w.shape[0]
    tiles = (W + _MFMA_TM - 1) // _MFMA_TM
    if tiles * _MFMA_TM > 1.15 * W:
        return False
    return _lib_supported(B, H, W, C, O)


def conv3x3_bias_relu(x, conv, live=None):
    """``relu(conv(x))`` for an ``nn.Conv2d`` that conv3x3_mfma_applies() accepted; ``live`` (uint8 per pixel of ONE
    image, ops/pyramid.CanvasLayout.live) zeroes the canvas gap pixels of the output."""
    return _Conv3x3BiasReLU.apply(x, conv.weight, conv.bias, live)


_TOWER = True    # two stacked conv + ReLU layers of the head canvas as one autograd node (False: one node per layer)


class _Conv3x3Tower2(torch.autograd.Function):
    """``relu(conv(relu(conv(x, w1) + b1), w2) + b2)`` on the head canvas (the two stacked ConvModules of an S2ANet tower,
    /root/reference/python/jdet/models/roi_heads/s2anet_head.py:130-170) as ONE node.  Forward: the two launches of
    _Conv3x3BiasReLU.  Backward, ordered by hand: the second layer's ReLU gate + bias gradient (one pass), its
    backward-data as csrc/conv3x3_mfma.hip on the flipped weights WITH THE FIRST LAYER'S gate and bias-gradient sums in
    the epilogue (rsdet_conv3x3_dgrad_gate_mfma_bf16) -- the first layer's own gate pass over the canvas and its fold
    launch are gone -- then the two weight gradients and the first layer's backward-data.  The gate reads the stored bf16
    c1 exactly as the per-layer pass does: identical gradients (tests/test_gpu_pyramid.py)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, live):
        xb = x.contiguous(memory_format=torch.channels_last)
        c1 = _mfma_conv(xb, w1, b1, live, True)
        c2 = _mfma_conv(c1, w2, b2, live, True)
        ctx.save_for_backward(xb, c1, c2, w1, w2)
        ctx.prep = (wprep.entry(w1, flip=True), wprep.entry(w2, flip=True))
        return c2

    @staticmethod
    def backward(ctx, gy):
        from .. import _lib
        xb, c1, c2, w1, w2 = ctx.saved_tensors
        lib = _lib.load()
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        g2, gb2 = _bias_relu_backward(gy, c2, ctx.needs_input_grad[4])
        B, O1, H, W = c1.shape
        g1 = torch.empty_like(c1)
        need_b1 = ctx.needs_input_grad[2]
        gb1 = torch.empty((O1,), dtype=torch.float32, device=c1.device) if need_b1 else None
        nb = lib.rsdet_conv3x3_dgrad_gate_ws_size(B, H, W, O1) if need_b1 else 0
        ws = torch.empty((nb,), dtype=torch.uint8, device=c1.device) if nb else None
        rc = lib.rsdet_conv3x3_dgrad_gate_mfma_bf16(_lib.ptr(g2), _lib.ptr(ctx.prep[1].tensor()), _lib.ptr(c1), B, H, W,
                                                    g2.shape[1], O1, _lib.ptr(g1), _lib.ptr(gb1), _lib.ptr(ws), nb,
                                                    _lib.stream_ptr())
        _lib.check(rc, "rsdet_conv3x3_dgrad_gate_mfma_bf16")
        gw2 = _mfma_wrw(g2, c1, w2.dtype) if ctx.needs_input_grad[3] else None
        if gw2 is None and ctx.needs_input_grad[3]:
            gw2 = torch.ops.aten.convolution_backward(g2, c1, w2, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                      (False, True, False))[1]
        del g2
        gx = _mfma_conv(g1, ctx.prep[0].tensor(), None, None, False) if ctx.needs_input_grad[0] else None
        gw1 = _mfma_wrw(g1, xb, w1.dtype) if ctx.needs_input_grad[1] else None
        if gw1 is None and ctx.needs_input_grad[1]:
            gw1 = torch.ops.aten.convolution_backward(g1, xb, w1, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                      (False, True, False))[1]
        return gx, gw1, gb1, gw2, gb2, None


def conv3x3_tower2_applies(x, conv_a, conv_b):
    """Both layers take the MFMA kernel (conv3x3_mfma_applies), their weights are bf16 channels_last Parameters (the
    prepared flipped operands of ops/weight_prep.py), biases fp32, square channel counts the backward's transposed
    problems tile."""
    if not (_TOWER and x.dtype == torch.bfloat16 and conv3x3_mfma_applies(x, conv_a)):
        return False
    for c in (conv_a, conv_b):
        if not (type(c) is torch.nn.Conv2d and tuple(c.weight.shape[2:]) == (3, 3) and tuple(c.stride) == (1, 1)
                and tuple(c.padding) == (1, 1) and tuple(c.dilation) == (1, 1) and c.groups == 1
                and c.padding_mode == 'zeros' and wprep.applies(c.weight) and c.bias is not None
                and c.bias.dtype == torch.float32 and c.weight.shape[0] % 64 == 0 and c.weight.shape[1] % 64 == 0):
            return False
    B, C, H, W = x.shape
    return (conv_a.weight.shape[1] == C and conv_b.weight.shape[1] == conv_a.weight.shape[0]
            and _lib_supported(B, H, W, conv_a.weight.shape[0], conv_b.weight.shape[0])
            and _lib_supported(B, H, W, conv_b.weight.shape[0], conv_a.weight.shape[0])
            and _lib_supported(B, H, W, conv_a.weight.shape[0], C))


def conv3x3_tower2(x, conv_a, conv_b, live=None):
    """The two-layer tower for a pair that conv3x3_tower2_applies() accepted."""
    return _Conv3x3Tower2.apply(x, conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias, live)


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
