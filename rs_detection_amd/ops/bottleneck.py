"""A whole identity Bottleneck of the bf16 trunk as ONE autograd node with a hand-ordered backward.

The reference block (/root/reference/python/jdet/models/backbones/resnet.py:57-93; forward :80-91) is

    y1 = relu(bn1(conv1(x)));  y2 = relu(bn2(conv2(y1)));  y3 = relu(bn3(conv3(y2)) + x)

with every BatchNorm in eval mode (norm_eval, :177-184).  Round 5's first stage put the BatchNorm tails of the two 1x1
convolutions into the GEMM's epilogue (ops/conv_bn.py).  With the block as one node the BACKWARD can be ordered by hand,
and three of its passes over the block's widest tensors disappear:

  * NO gradient of a convolution's raw output is ever written: every gated gradient (gz = gy [y3 > 0], [y2 > 0] ...,
    [y1 > 0] ...) is the gradient of a BatchNorm's OUTPUT, and that BatchNorm's scale s = gamma / sqrt(var + eps) rides in
    the prepared weights of the backward-data step that reads it (ops/weight_prep.py: (W3 s3)^T, flipped W2 s2,
    (W1 s1)^T) and in the fold of the weight gradient that reads it (gw = s U).  gz of the block's output is ALSO the
    identity branch's gradient, so one tensor serves three consumers;
  * bn2's gate (y2 > 0) and bias-gradient sums are the EPILOGUE of conv3's backward-data GEMM (csrc/gemm1x1_mfma.hip,
    mode 2): conv3's input gradient never reaches memory ungated, conv2's raw output and its ReLU bit mask are not kept
    for the backward at all;
  * the sum of the two gradients that reach x (through conv1 and through the identity) is the epilogue of conv1's
    backward-data GEMM (mode 3) instead of an elementwise pass over three tensors of the block's widest shape;
  * the three scale gradients come from the weight-gradient folds: sum_p gz[p, o] conv(x)[p, o] = sum_k W[o, k] U[o, k]
    with U the unscaled fp32 weight gradient, so grad_gamma = (rowdot - mean grad_beta) / sqrt(var + eps) needs neither the
    convolution output nor xhat = (y - beta) / gamma recovered from a bf16 output (what round 5 did: the rounding of y is
    amplified by 1 / gamma -- wrong sign below gamma ~ 1e-3, nothing at gamma = 0).  Exact for any gamma.

Same sums as the per-operator route; the scales folded into bf16 weights are rounded once more than there
(tests/test_gpu_bottleneck.py pins both against the fp32 composite, small and zero gammas included).

Applies to: identity blocks (no downsample, stride 1, groups 1) on CUDA with bf16 channels_last activations, bf16
weights (Runner(bf16_params=True)), eval-mode affine BatchNorms with fp32 parameters, every parameter trainable, channel
counts the kernels tile -- the 10 of 13 trainable blocks of ResNet-50, 27 of 30 of ResNet-101.  Everything else runs
the per-operator forward of models/backbones/resnet.py."""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib
from . import weight_prep as wprep
from .bn_act import _memo
from .conv1x1 import _wrw_split_k
from .conv3x3 import _mfma_wrw

_ON = True      # False: the per-operator route (what this node is tested against)
_PTR3, _INT3, _FLT3 = ctypes.c_void_p * 3, ctypes.c_int * 3, ctypes.c_float * 3


def _cl_empty(B, C, H, W, device):
    return torch.empty((B, C, H, W), dtype=torch.bfloat16, device=device, memory_format=torch.channels_last)


def _conv_bn_fwd(lib, x, w, bn, residual):
    gamma, beta, mean, var, eps = bn
    B, C, H, W = x.shape
    O = w.shape[0]
    y = _cl_empty(B, O, H, W, x.device)
    rc = lib.rsdet_conv1x1_bn_act_fwd_bf16(_lib.ptr(x), _lib.ptr(w), B * H * W, O, C, _lib.ptr(mean), _lib.ptr(var),
                                           _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.ptr(residual), 1, _lib.ptr(y),
                                           _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv1x1_bn_act_fwd_bf16")
    return y


class _Bottleneck(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, w3, ga1, be1, ga2, be2, ga3, be3, stats, pad):
        lib = _lib.load()
        (m1, v1, e1), (m2, v2, e2), (m3, v3, e3) = stats
        y1 = _conv_bn_fwd(lib, x, w1, (ga1, be1, m1, v1, e1), None)
        c2 = F.conv2d(y1, w2, None, 1, pad, pad)
        B, C1, H, W = c2.shape
        y2 = torch.empty_like(c2)
        rc = lib.rsdet_bn_act_forward_nhwc_bf16(_lib.ptr(c2), None, _lib.ptr(m2), _lib.ptr(v2), _lib.ptr(ga2),
                                                _lib.ptr(be2), e2, B, C1, H * W, 1, _lib.ptr(y2), _lib.stream_ptr())
        _lib.check(rc, "rsdet_bn_act_forward_nhwc_bf16")
        del c2
        y3 = _conv_bn_fwd(lib, y2, w3, (ga3, be3, m3, v3, e3), x)
        ctx.save_for_backward(x, y1, y2, y3, w1, w2, w3, ga1, ga2, ga3)
        ctx.stats, ctx.pad = stats, pad
        # the operands of the backward that are functions of the weights (and BatchNorm scales) alone: one launch per
        # optimizer step for all blocks (ops/weight_prep.py)
        ctx.prep = (wprep.entry(w1, bn=(v1, ga1, e1)), wprep.entry(w2, bn=(v2, ga2, e2), flip=True) if pad == 1 else None,
                    wprep.entry(w3, bn=(v3, ga3, e3)))
        return y3

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, y1, y2, y3, w1, w2, w3, ga1, ga2, ga3 = ctx.saved_tensors
        (m1, v1, e1), (m2, v2, e2), (m3, v3, e3) = ctx.stats
        B, C0, H, W = x.shape
        C1 = w1.shape[0]
        P, HW = B * H * W, H * W
        dev = x.device
        st = _lib.stream_ptr()
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        f32 = dict(dtype=torch.float32, device=dev)

        def ws_for(fn, *shape):
            nb = _memo(fn, *shape)
            return torch.empty((nb,), dtype=torch.uint8, device=dev), nb

        # the beta / gamma gradients of the three BatchNorms: per-slice sum tables + the folds' row dots, ONE launch at the end
        gsum = torch.empty((2, 2 * C1 + C0), **f32)            # row 0: the gamma gradients (bn1 | bn2 | bn3), row 1: beta
        gga1, gga2, gga3 = gsum[0, :C1], gsum[0, C1:2 * C1], gsum[0, 2 * C1:]
        gbe1, gbe2, gbe3 = gsum[1, :C1], gsum[1, C1:2 * C1], gsum[1, 2 * C1:]
        # ---- 1: through relu(. + x): gz = gy [y3 > 0] (the gradient of bn3's output AND the identity's), its channel sums
        gz = torch.empty_like(gy)
        ws3, nb = ws_for("rsdet_bn_act_backward_nhwc_ws_size", B, C0, HW)
        rc = lib.rsdet_bn_gate_sums_nhwc_bf16(_lib.ptr(gy), _lib.ptr(y3), B, C0, HW, 1, _lib.ptr(gz), _lib.ptr(ws3), nb, st)
        _lib.check(rc, "rsdet_bn_gate_sums_nhwc_bf16")
        gz2 = gz.permute(0, 2, 3, 1).reshape(P, C0)
        # ---- 2: conv3's weight gradient from gz, bn3's scale applied in the fold; the fold's row dots -> bn3's gamma
        gw3, d3 = _wrw_split_k(gz2, y2, w3, rowscale=(v3, ga3, e3), rowdot=True)
        # ---- 3: conv3's backward-data with bn3's scale in the weights, bn2's gate + sums in the epilogue
        g2 = _cl_empty(B, C1, H, W, dev)                       # the gated gradient of bn2's OUTPUT
        ws2, nb = ws_for("rsdet_conv1x1_dgrad_ws_size", P, C1, C0)
        rc = lib.rsdet_conv1x1_dgrad_bf16(_lib.ptr(gz), _lib.ptr(ctx.prep[2].tensor()), P, C1, C0, 2, _lib.ptr(y2), None,
                                          _lib.ptr(ws2), nb, _lib.ptr(g2), st)
        _lib.check(rc, "rsdet_conv1x1_dgrad_bf16")
        # ---- 4: conv2 (3x3), bn2's scale in its operands: backward-data through the forward solver on the flipped,
        #         scaled weights; our split-K weight gradient with the scale (and the row dots) in its fold
        pad = ctx.pad
        bn2 = (v2, ga2, e2)
        gw2 = d2 = None
        if pad == 1:
            g1 = F.conv2d(g2, ctx.prep[1].tensor(), None, 1, 1)
            if C1 % 128 == 0:
                r = _mfma_wrw(g2, y1, w2.dtype, rowscale=bn2, weight=w2)
                if r is not None:
                    gw2, d2 = r
            if gw2 is None:
                u2 = torch.ops.aten.convolution_backward(g2, y1, w2, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
                gw2, d2 = _scale_rows(u2, w2, bn2)
        else:
            sc2 = _scale_of(bn2)
            w2s = (w2.float() * sc2[:, None, None, None]).to(w2.dtype).contiguous(memory_format=torch.channels_last)
            g1, u2, _ = torch.ops.aten.convolution_backward(g2, y1, w2s, None, (1, 1), (pad, pad), (pad, pad), False,
                                                            (0, 0), 1, (True, True, False))
            g1 = g1.contiguous(memory_format=torch.channels_last)
            gw2, d2 = _scale_rows(u2, w2, bn2)
        del g2
        # ---- 5: bn1's gate from y1 (the gradient of bn1's OUTPUT), its channel sums
        gz1 = torch.empty_like(g1)
        ws1, nb = ws_for("rsdet_bn_act_backward_nhwc_ws_size", B, C1, HW)
        rc = lib.rsdet_bn_gate_sums_nhwc_bf16(_lib.ptr(g1), _lib.ptr(y1), B, C1, HW, 1, _lib.ptr(gz1), _lib.ptr(ws1), nb, st)
        _lib.check(rc, "rsdet_bn_gate_sums_nhwc_bf16")
        del g1
        # ---- 6: conv1's weight gradient, bn1's scale in the fold
        gw1, d1 = _wrw_split_k(gz1.permute(0, 2, 3, 1).reshape(P, C1), x, w1, rowscale=(v1, ga1, e1), rowdot=True)
        # ---- the three BatchNorms' parameter gradients as one launch
        S = (_memo("rsdet_bn_gate_sums_nhwc_slices", B, C1, HW), _memo("rsdet_conv1x1_dgrad_slices", P, C1, C0),
             _memo("rsdet_bn_gate_sums_nhwc_slices", B, C0, HW))
        rc = lib.rsdet_bn_affine_grads_finish_multi_f32(
            3, _PTR3(ws1.data_ptr(), ws2.data_ptr(), ws3.data_ptr()), _INT3(C1, C1, C0), _INT3(*S),
            _PTR3(d1.data_ptr(), d2.data_ptr(), d3.data_ptr()), _PTR3(m1.data_ptr(), m2.data_ptr(), m3.data_ptr()),
            _PTR3(v1.data_ptr(), v2.data_ptr(), v3.data_ptr()), _FLT3(e1, e2, e3),
            _PTR3(gga1.data_ptr(), gga2.data_ptr(), gga3.data_ptr()), _PTR3(gbe1.data_ptr(), gbe2.data_ptr(), gbe3.data_ptr()),
            st)
        _lib.check(rc, "rsdet_bn_affine_grads_finish_multi_f32")
        # ---- 7: conv1's backward-data (bn1's scale in the weights) + the identity branch's gz
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(gz)
            rc = lib.rsdet_conv1x1_dgrad_bf16(_lib.ptr(gz1), _lib.ptr(ctx.prep[0].tensor()), P, C0, C1, 3, _lib.ptr(gz),
                                              None, None, 0, _lib.ptr(gx), st)
            _lib.check(rc, "rsdet_conv1x1_dgrad_bf16")
        return gx, gw1, gw2, gw3, gga1, gbe1, gga2, gbe2, gga3, gbe3, None, None


def _scale_of(bn):
    var, gamma, eps = bn
    return torch.rsqrt(var + eps) * gamma


def _scale_rows(u, w, bn):
    """(s u, rowdot) from an UNSCALED weight gradient u a library kernel produced (bf16): the layers our split-K kernels do
    not tile (64-channel and dilated conv2)."""
    uf = u.float()
    d = (uf * w.float()).sum((1, 2, 3))
    gw = (uf * _scale_of(bn)[:, None, None, None]).to(w.dtype)
    return gw.contiguous(memory_format=torch.channels_last), d


def _bn_ok(bn):
    return (type(bn) is torch.nn.BatchNorm2d and not bn.training and bn.running_mean is not None
            and bn.running_mean.dtype == torch.float32 and bn.weight is not None and bn.weight.dtype == torch.float32
            and bn.weight.requires_grad and bn.bias.requires_grad)


def _conv_ok(conv, k):
    return (type(conv) is torch.nn.Conv2d and conv.kernel_size == (k, k) and conv.stride == (1, 1) and conv.groups == 1
            and conv.bias is None and conv.padding_mode == 'zeros' and conv.weight.requires_grad
            and wprep.applies(conv.weight))


def bottleneck_applies(block, x):
    """Does the one-node form take this models/backbones/resnet.py Bottleneck on this input?  (module docstring)"""
    if not (_ON and block.downsample is None and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and x.dtype == torch.bfloat16 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    c1, c2, c3 = block.conv1, block.conv2, block.conv3
    if not (_conv_ok(c1, 1) and _conv_ok(c2, 3) and _conv_ok(c3, 1) and _bn_ok(block.bn1) and _bn_ok(block.bn2)
            and _bn_ok(block.bn3)):
        return False
    if not (c1.padding == (0, 0) and c3.padding == (0, 0) and c2.padding == c2.dilation and c2.padding[0] == c2.padding[1]
            and c1.dilation == (1, 1) and c3.dilation == (1, 1)):
        return False
    B, C0, H, W = x.shape
    C1 = c1.out_channels
    if not (c3.out_channels == C0 and c2.in_channels == C1 and c2.out_channels == C1 and c3.in_channels == C1):
        return False
    P = B * H * W

    def nhwc8(c):           # the eight-channel BatchNorm backward pass: C / 8 divides 256
        return c % 8 == 0 and c <= 2048 and 256 % (c // 8) == 0
    return (nhwc8(C0) and nhwc8(C1) and bool(_memo("rsdet_gemm1x1_mfma_supported", P, C1, C0))
            and bool(_memo("rsdet_gemm1x1_mfma_supported", P, C0, C1)))


def bottleneck(block, x):
    """``block(x)`` for a Bottleneck that bottleneck_applies() accepted."""
    bn1, bn2, bn3 = block.bn1, block.bn2, block.bn3
    stats = ((bn1.running_mean, bn1.running_var, float(bn1.eps)), (bn2.running_mean, bn2.running_var, float(bn2.eps)),
             (bn3.running_mean, bn3.running_var, float(bn3.eps)))
    return _Bottleneck.apply(x, block.conv1.weight, block.conv2.weight, block.conv3.weight, bn1.weight, bn1.bias,
                             bn2.weight, bn2.bias, bn3.weight, bn3.bias, stats, int(block.conv2.padding[0]))
