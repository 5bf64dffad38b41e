"""1x1 convolution + eval-mode BatchNorm + residual add + ReLU of the bf16 trunk as ONE launch forward
(csrc/gemm1x1_mfma.hip) -- host side.

The Bottleneck of the reference (/root/reference/python/jdet/models/backbones/resnet.py:57-93) runs
``relu(bn(conv(x)) [+ identity])`` with every BatchNorm in eval mode (norm_eval, :177-184).  Rounds 3-4 ran the 1x1
convolutions as library GEMMs on (positions, channels) views (ops/conv1x1.py) and the BatchNorm tail as one fused pass
over the result (ops/bn_act.py); here the tail sits in the GEMM's epilogue and the convolution's raw output never
reaches memory: one launch and one autograd node instead of two, 0.86 ms of `bn_act` forward passes less per bf16 step.

Backward (one node): one gate pass gz = gy [y > 0] with the bias-gradient sums (csrc/bn_act.hip,
rsdet_bn_gate_sums_nhwc_bf16); the BatchNorm's scale s = gamma / sqrt(var + eps) then rides in the WEIGHTS of the
backward-data GEMM (gx = gz (W s)) and in the fold of the split-K weight gradient (gw = s U, U = gz^T x in fp32), and the
scale gradient comes from that fold too: sum_p gz conv(x) = sum_c W[o, c] U[o, c], so
grad_gamma = (rowdot - mean grad_beta) / sqrt(var + eps) -- the convolution output the forward computed, never stored
and never recovered from the bf16 output (round 5 divided (y - beta) by gamma: unstable for small gamma, lost for
gamma = 0).
Applies to: CUDA, bf16 activations AND bf16 weights (``Runner(bf16_params=True)``), channels_last, 1x1 / stride 1 / no
bias / groups 1, C % 64 == 0, O % 32 == 0, BatchNorm in eval mode with fp32 parameters.  Anything else takes
``bn_act(conv1x1(conv, x), bn, residual, relu)`` -- the same function in two launches."""
import torch

import ctypes

from .. import _lib
from . import weight_prep as wprep
from .bn_act import _memo, bn_act
from .conv1x1 import _wrw_split_k, conv1x1

_P1, _I1, _F1 = ctypes.c_void_p * 1, ctypes.c_int * 1, ctypes.c_float * 1

_ON = True      # False: always the two-launch form (what the fused form is tested against)


class _Conv1x1BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, mean, var, eps, residual, relu):
        lib = _lib.load()
        B, C, H, W = x.shape
        O = w.shape[0]
        P = B * H * W
        y = torch.empty((B, O, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        rc = lib.rsdet_conv1x1_bn_act_fwd_bf16(_lib.ptr(x), _lib.ptr(w), P, O, C, _lib.ptr(mean), _lib.ptr(var),
                                               _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(residual),
                                               int(relu), _lib.ptr(y), _lib.stream_ptr())
        _lib.check(rc, "rsdet_conv1x1_bn_act_fwd_bf16")
        ctx.save_for_backward(x, w, y, gamma, mean, var)
        ctx.eps, ctx.relu, ctx.has_res = float(eps), bool(relu), residual is not None
        # the backward-data operand (W s)^T: refreshed with all the others once per optimizer step (ops/weight_prep.py)
        ctx.prep = wprep.entry(w, bn=(var, gamma, float(eps))) if wprep.applies(w) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w, y, gamma, mean, var = ctx.saved_tensors
        B, C, H, W = x.shape
        O = w.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_g = gamma is not None and ctx.needs_input_grad[2]
        need_b = ctx.needs_input_grad[3]
        need_res = ctx.has_res and ctx.needs_input_grad[7]
        # ---- through relu: gz = the gradient of the BatchNorm's OUTPUT (and of the identity branch), its channel sums
        sums = need_g or need_b
        gz = torch.empty_like(gy) if ctx.relu else gy
        ws_bytes = _memo("rsdet_bn_act_backward_nhwc_ws_size", B, O, H * W) if sums else 0
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gy.device) if ws_bytes else None
        if ctx.relu or sums:
            rc = lib.rsdet_bn_gate_sums_nhwc_bf16(_lib.ptr(gy), _lib.ptr(y), B, O, H * W, int(ctx.relu),
                                                  _lib.ptr(gz) if ctx.relu else None, _lib.ptr(ws), ws_bytes,
                                                  _lib.stream_ptr())
            _lib.check(rc, "rsdet_bn_gate_sums_nhwc_bf16")
        gx = gw = gg = gb = d = None
        gz2 = gz.permute(0, 2, 3, 1).reshape(-1, O)
        scale = (var, gamma, ctx.eps)
        if need_w or need_g:
            gw, d = _wrw_split_k(gz2, x, w, rowscale=scale, rowdot=True)
        if need_x:
            wt = ctx.prep.tensor() if ctx.prep is not None else _scaled_t(w, scale)        # (C, O) = (W s)^T
            gx = torch.mm(gz2, wt.t()).view(B, H, W, C).permute(0, 3, 1, 2)
        if sums:
            gb = torch.empty((O,), dtype=torch.float32, device=gy.device)
            gg = torch.empty((O,), dtype=torch.float32, device=gy.device) if need_g else None
            S = _memo("rsdet_bn_gate_sums_nhwc_slices", B, O, H * W)
            rc = lib.rsdet_bn_affine_grads_finish_multi_f32(
                1, _P1(ws.data_ptr()), _I1(O), _I1(S), _P1(d.data_ptr() if need_g else None), _P1(mean.data_ptr()),
                _P1(var.data_ptr()), _F1(ctx.eps), _P1(gg.data_ptr() if need_g else None), _P1(gb.data_ptr()),
                _lib.stream_ptr())
            _lib.check(rc, "rsdet_bn_affine_grads_finish_multi_f32")
        return gx, (gw if need_w else None), gg, (gb if need_b else None), None, None, None, \
            (gz if need_res else None), None


def _scaled_t(w, scale):
    var, gamma, eps = scale
    sc = torch.rsqrt(var + eps) * (1.0 if gamma is None else gamma)
    O, C = w.shape[0], w.shape[1]
    return (w.reshape(O, C).float() * sc[:, None]).to(w.dtype).t().contiguous()


def conv_bn_act_applies(conv, bn, x, residual):
    if not (_ON and type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.weight.dtype == torch.bfloat16
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    if bn.training or bn.running_mean is None or bn.running_mean.dtype != torch.float32 or \
            (bn.weight is not None and bn.weight.dtype != torch.float32):
        return False
    if residual is not None and not (residual.dtype == x.dtype and not residual.is_contiguous()
                                     and residual.is_contiguous(memory_format=torch.channels_last)
                                     and tuple(residual.shape) == (x.shape[0], conv.out_channels, x.shape[2], x.shape[3])):
        return False
    B, C, H, W = x.shape
    O = conv.out_channels
    # (the backward's eight-channel pass wants O / 8 to divide 256: every ResNet width from 32 to 2 048)
    return O % 8 == 0 and O <= 2048 and 256 % (O // 8) == 0 and \
        bool(_memo("rsdet_gemm1x1_mfma_supported", B * H * W, O, C))


def conv_bn_act(conv, bn, x, residual=None, relu=True, fork=False):
    """``relu(bn(conv(x)) + residual)`` for an nn.Conv2d + nn.BatchNorm2d pair: one launch where the fused kernel applies
    (module docstring), ``bn_act(conv1x1(conv, x), bn, residual, relu)`` otherwise (``fork``: ops/bn_act.Forked)."""
    if conv_bn_act_applies(conv, bn, x, residual):
        return _Conv1x1BNAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                   residual, relu)
    return bn_act(conv1x1(conv, x), bn, residual=residual, relu=relu, fork=fork)
