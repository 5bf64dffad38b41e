"""jdet.ops.orn on MI355X: active_rotating_filter, ORConv2d, RotationInvariantPooling.

Mirror of /root/reference/python/jdet/ops/orn.py:543-555 (autograd function),
:595-617 (RotationInvariantPooling), :620-705 (ORConv2d).  ARF kernels:
rs_detection_amd/csrc/arf.hip; the convolution itself is MIOpen via torch.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

__all__ = ["ORConv2d", "RotationInvariantPooling", "active_rotating_filter", "arf_forward", "arf_backward"]


def arf_forward(weight, indices):
    """orn.py:260-269: weight (O,I,nOri,kH,kW) f32, indices (nOri,kH,kW,nRot) u8 -> (O*nRot, I*nOri, kH, kW)."""
    assert weight.dim() == 5, "only supports a batch of ARFs."
    assert weight.dtype == torch.float32 and indices.dtype == torch.uint8
    _lib.require_cuda_f32(weight)
    lib = _lib.load()
    w, idx = weight.contiguous(), indices.contiguous()
    O, I, nOri, kH, kW = w.shape
    nRot = idx.shape[3]
    out = torch.empty((O * nRot, I * nOri, kH, kW), dtype=w.dtype, device=w.device)
    rc = lib.rsdet_arf_forward_f32(_lib.ptr(w), _lib.ptr(idx), O, I, nOri, kH, kW, nRot, _lib.ptr(out),
                                   _lib.stream_ptr())
    _lib.check(rc, "rsdet_arf_forward_f32")
    return out


def arf_backward(indices, grad_output):
    """orn.py:271-280."""
    assert indices.dim() == 4 and indices.dtype == torch.uint8 and grad_output.dtype == torch.float32
    _lib.require_cuda_f32(grad_output)
    lib = _lib.load()
    idx, go = indices.contiguous(), grad_output.contiguous()
    nOri, kH, kW, nRot = idx.shape
    O, I = go.shape[0] // nRot, go.shape[1] // nOri
    gw = torch.empty((O, I, nOri, kH, kW), dtype=go.dtype, device=go.device)
    rc = lib.rsdet_arf_backward_f32(_lib.ptr(idx), _lib.ptr(go), O, I, nOri, kH, kW, nRot, _lib.ptr(gw),
                                    _lib.stream_ptr())
    _lib.check(rc, "rsdet_arf_backward_f32")
    return gw


class _ActiveRotatingFilter(torch.autograd.Function):
    """orn.py:543-555."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, indices):
        indices = indices.to(torch.uint8)
        ctx.save_for_backward(indices)
        return arf_forward(input, indices)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return arf_backward(indices, grad_output), None


active_rotating_filter = _ActiveRotatingFilter.apply


def rie_forward(feature, nOrientation):
    """orn.py:516-530: feature (N, C, 1, 1) -> (mainDirection (N, C/nOri) uint8, aligned like feature)."""
    assert feature.dim() == 4, "only supports a batch of RIEs."
    assert feature.size(2) == 1 and feature.size(3) == 1, "mH x mW should be 1x1."
    _lib.require_cuda_f32(feature)
    f = feature.contiguous()
    N, C = f.shape[:2]
    nF = C // nOrientation
    d = torch.empty((N, nF), dtype=torch.uint8, device=f.device)
    out = torch.empty_like(f)
    rc = _lib.load().rsdet_rie_forward_f32(_lib.ptr(f), N, nF, int(nOrientation), _lib.ptr(d), _lib.ptr(out),
                                           _lib.stream_ptr())
    _lib.check(rc, "rsdet_rie_forward_f32")
    return d, out


def rie_backward(mainDirection, grad_output, nOrientation):
    """orn.py:533-540."""
    _lib.require_cuda_f32(grad_output)
    g = grad_output.contiguous()
    N, nF = mainDirection.shape
    gi = torch.empty_like(g)
    rc = _lib.load().rsdet_rie_backward_f32(_lib.ptr(mainDirection.contiguous()), _lib.ptr(g), N, nF, int(nOrientation),
                                            _lib.ptr(gi), _lib.stream_ptr())
    _lib.check(rc, "rsdet_rie_backward_f32")
    return gi


class _RotationInvariantEncoding(torch.autograd.Function):
    """orn.py:557-567."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, input, nOrientation):
        d, out = rie_forward(input, nOrientation)
        ctx.nOrientation = nOrientation
        ctx.save_for_backward(d)
        ctx.mark_non_differentiable(d)
        return out, d

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output, _grad_direction=None):
        (d,) = ctx.saved_tensors
        return rie_backward(d, grad_output, ctx.nOrientation), None


rotation_invariant_encoding = _RotationInvariantEncoding.apply


class RotationInvariantEncoding(nn.Module):
    """orn.py:582-593."""

    def __init__(self, nOrientation, return_direction=False):
        super().__init__()
        self.nOrientation, self.return_direction = nOrientation, return_direction

    def forward(self, input):
        output, d = rotation_invariant_encoding(input, self.nOrientation)
        return (output, d) if self.return_direction else output


class RotationInvariantPooling(nn.Module):
    """orn.py:595-617: max over the orientation axis.  The 1x1 conv + BN members are
    parameters of the reference module that its forward bypasses (:615-616); they are
    kept (frozen) so checkpoints keep the same keys."""

    def __init__(self, nInputPlane, nOrientation=8):
        super().__init__()
        self.nInputPlane = nInputPlane
        self.nOrientation = nOrientation
        hidden = int(nInputPlane / nOrientation)
        self.conv = nn.Sequential(nn.Conv2d(hidden, nInputPlane, 1, 1), nn.BatchNorm2d(nInputPlane))
        for p in self.conv.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        N, c, h, w = x.size()
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and c % self.nOrientation == 0 and \
                (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            return _OriMaxPool.apply(x, self.nOrientation)
        return x.view(N, -1, self.nOrientation, h, w).amax(dim=2)


class _OriMaxPool(torch.autograd.Function):
    """max over the orientation channels as one pass each way (csrc/arf.hip: ori_maxpool_*), NCHW or channels_last,
    fp32 or bf16; gradients shared equally by tied maxima like the torch.amax it replaces."""

    @staticmethod
    def forward(ctx, x, nori):
        lib = _lib.load()
        N, C, H, W = x.shape
        nhwc = not x.is_contiguous()
        y = torch.empty((N, C // nori, H, W), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        _lib.check(lib.rsdet_ori_maxpool_forward(_lib.ptr(x), int(x.dtype == torch.bfloat16), N, C // nori, nori, H * W,
                                                 int(nhwc), _lib.ptr(y), _lib.stream_ptr()), "rsdet_ori_maxpool_forward")
        ctx.save_for_backward(x)
        ctx.nori, ctx.nhwc = nori, nhwc
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last if ctx.nhwc else torch.contiguous_format).to(x.dtype)
        gx = torch.empty_like(x)
        _lib.check(lib.rsdet_ori_maxpool_backward(_lib.ptr(x), _lib.ptr(gy), int(x.dtype == torch.bfloat16), N,
                                                  C // ctx.nori, ctx.nori, H * W, int(ctx.nhwc), _lib.ptr(gx),
                                                  _lib.stream_ptr()), "rsdet_ori_maxpool_backward")
        return gx, None


_KERNEL_INDICES = {
    1: {a: (1,) for a in (0, 45, 90, 135, 180, 225, 270, 315)},
    3: {  # orn.py:655-664: the eight 45-degree permutations of a 3x3 stencil
        0: (1, 2, 3, 4, 5, 6, 7, 8, 9),
        45: (2, 3, 6, 1, 5, 9, 4, 7, 8),
        90: (3, 6, 9, 2, 5, 8, 1, 4, 7),
        135: (6, 9, 8, 3, 5, 7, 2, 1, 4),
        180: (9, 8, 7, 6, 5, 4, 3, 2, 1),
        225: (8, 7, 4, 9, 5, 1, 6, 3, 2),
        270: (7, 4, 1, 8, 5, 2, 9, 6, 3),
        315: (4, 1, 2, 7, 5, 3, 8, 9, 6),
    },
}


def arf_indices(nOrientation, nRotation, kernel_size):
    """orn.py:644-678 -> uint8 (nOri, kH, kW, nRot), 1-based."""
    kH, kW = kernel_size
    d_ori = 360 / nOrientation
    d_rot = 360 / nRotation
    idx = torch.zeros((nOrientation * kH * kW, nRotation), dtype=torch.uint8)
    for i in range(nOrientation):
        for j in range(kH * kW):
            for k in range(nRotation):
                angle = d_rot * k
                layer = (i + math.floor(angle / d_ori)) % nOrientation
                kernel = _KERNEL_INDICES[kW][angle][j]
                idx[i * kH * kW + j, k] = int(layer * kH * kW + kernel)
    return idx.view(nOrientation, kH, kW, nRotation)


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class ORConv2d(nn.Conv2d):
    """orn.py:620-705.  weight (out, in, nOri, kH, kW); bias (out*nRot,)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, arf_config=None, stride=1, padding=0, dilation=1,
                 groups=1, bias=True):
        self.nOrientation, self.nRotation = _pair(arf_config)
        assert (math.log(self.nOrientation) + 1e-5) % math.log(2) < 1e-3, 'invalid nOrientation {}'.format(self.nOrientation)
        assert (math.log(self.nRotation) + 1e-5) % math.log(2) < 1e-3, 'invalid nRotation {}'.format(self.nRotation)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.register_buffer("indices", arf_indices(self.nOrientation, self.nRotation, self.kernel_size))
        self.weight = nn.Parameter(torch.zeros((out_channels, in_channels, self.nOrientation, *self.kernel_size)))
        if bias:
            self.bias = nn.Parameter(torch.zeros((out_channels * self.nRotation,)))
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight.dim() != 5:  # called by nn.Conv2d.__init__ before the ARF weight exists
            return super().reset_parameters()
        n = self.in_channels * self.nOrientation
        for k in self.kernel_size:
            n *= k
        nn.init.normal_(self.weight, 0, math.sqrt(2.0 / n))

    def rotate_arf(self):
        return active_rotating_filter(self.weight, self.indices)

    def forward(self, input):
        w = self.rotate_arf()
        from .conv3x3 import conv3x3_applies, _Conv3x3Same
        if conv3x3_applies(input, w, self.stride, self.padding, self.dilation, self.groups):
            return _Conv3x3Same.apply(input, w, self.bias)     # backward-data through the forward solver
        return F.conv2d(input, w, self.bias, self.stride, self.padding, self.dilation, self.groups)
