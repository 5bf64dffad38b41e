"""Depthwise 2-D convolution of the VAN backbone on MI355X (csrc/dwconv.hip).

`DepthwiseConv2d` IS an `nn.Conv2d(dim, dim, k, padding=dilation*(k-1)//2, groups=dim, dilation=dilation)` -- same
parameters, names, shapes and initialisation (checkpoints load unchanged; reference call sites:
/root/reference/python/jdet/models/backbones/van.py:32,56,57) -- whose forward / backward run the LDS-tiled stencil
kernels when the input is a CUDA float32 tensor and the geometry is one the kernels cover; every other case is
torch's own convolution (MIOpen), e.g. the CPU unit tests of the model.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

__all__ = ["DepthwiseConv2d", "dwconv2d"]

_COVERED = {(3, 1), (5, 1), (7, 3)}


class _DWConv2d(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, dilation, in_bias=None):
        _lib.require_cuda_f32(x, weight, bias, in_bias)
        lib = _lib.load()
        x, weight = x.contiguous(), weight.contiguous()
        N, C, H, W = x.shape
        K = weight.shape[-1]
        assert weight.shape == (C, 1, K, K)
        y = torch.empty_like(x)
        rc = lib.rsdet_dwconv2d_forward_f32(_lib.ptr(x), _lib.ptr(in_bias.contiguous()) if in_bias is not None else None,
                                            _lib.ptr(weight), _lib.ptr(bias.contiguous()) if bias is not None else None,
                                            N, C, H, W, K, int(dilation), _lib.ptr(y), _lib.stream_ptr())
        _lib.check(rc, "rsdet_dwconv2d_forward_f32")
        ctx.save_for_backward(x, weight, in_bias)
        ctx.cfg = (int(dilation), bias is not None)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        x, weight, in_bias = ctx.saved_tensors
        dilation, has_bias = ctx.cfg
        lib = _lib.load()
        gy = gy.contiguous()
        N, C, H, W = x.shape
        K = weight.shape[-1]
        gx = gw = gb = gib = None
        need_ib = in_bias is not None and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[0] or need_ib:
            gx = torch.empty_like(x)
            gib = torch.empty((C,), dtype=x.dtype, device=x.device) if need_ib else None
            ws_bytes = lib.rsdet_dwconv2d_backward_data_ws_size(N, C, H, W) if need_ib else 0
            ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=x.device) if need_ib else None
            rc = lib.rsdet_dwconv2d_backward_data_f32(_lib.ptr(gy), _lib.ptr(weight), N, C, H, W, K, dilation,
                                                      _lib.ptr(gx), _lib.ptr(gib), _lib.ptr(ws), ws_bytes,
                                                      _lib.stream_ptr())
            _lib.check(rc, "rsdet_dwconv2d_backward_data_f32")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty((C,), dtype=x.dtype, device=x.device) if has_bias else None
            ws_bytes = lib.rsdet_dwconv2d_backward_weight_ws_size(N, C, H, W, K)
            ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=x.device)
            rc = lib.rsdet_dwconv2d_backward_weight_f32(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(in_bias), N, C, H, W, K,
                                                        dilation, _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ws_bytes,
                                                        _lib.stream_ptr())
            _lib.check(rc, "rsdet_dwconv2d_backward_weight_f32")
        return (gx if ctx.needs_input_grad[0] else None), gw, gb, None, gib


def dwconv2d(x, weight, bias=None, dilation=1, in_bias=None):
    """Depthwise "same" convolution, stride 1: x (N,C,H,W), weight (C,1,K,K); (K, dilation) in {(3,1),(5,1),(7,3)}.
    ``in_bias`` (C): computes conv(x + in_bias[None, :, None, None]) -- the bias of the 1x1 convolution that produced x,
    folded into this kernel's load (its gradient comes back as the gradient of ``in_bias``)."""
    return _DWConv2d.apply(x, weight, bias, dilation, in_bias)


class DepthwiseConv2d(nn.Conv2d):
    def __init__(self, dim, kernel_size, padding=0, dilation=1, bias=True, stride=1):
        super().__init__(dim, dim, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=dim, bias=bias)

    def _covered(self, x):
        k, d = self.kernel_size[0], self.dilation[0]
        return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and self.weight.dtype == torch.float32
                and self.kernel_size[0] == self.kernel_size[1] and self.dilation[0] == self.dilation[1]
                and (k, d) in _COVERED and self.stride == (1, 1) and self.padding == (d * (k - 1) // 2,) * 2
                and self.padding_mode == 'zeros' and x.shape[2] * x.shape[3] > 0)

    def forward(self, x, in_bias=None):
        """``in_bias``: per-channel constant owed to ``x`` by its producer (see dwconv2d); added here, in the kernel's
        load on the covered path, as a plain broadcast add otherwise."""
        if self._covered(x):
            return dwconv2d(x, self.weight, self.bias, self.dilation[0], in_bias)
        if in_bias is not None:
            x = x + in_bias.to(x.dtype)[None, :, None, None]
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
