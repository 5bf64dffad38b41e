"""LayerNorm over the channels of an NCHW map (csrc/van_ops.hip: ln_chan_*) -- host side.

Reference: /root/reference/python/jdet/models/backbones/van.py:303-306 -- every VAN stage ends with
``x = norm(x.flatten(2).transpose(1, 2)); x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2)``.  As tensor operations that is a
strided copy, the library LayerNorm and a strided copy back (and the same again in the backward): ~1.0 ms per Oriented R-CNN
step for four calls.  One launch each way here, the map read once and written once in the forward."""
import torch

from .. import _lib

_ON = True


class _ChanLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rc = lib.rsdet_chan_layernorm_forward_f32(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), N, C, H * W, float(eps),
                                                  _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), _lib.stream_ptr())
        _lib.check(rc, "rsdet_chan_layernorm_forward_f32")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        gb = torch.empty_like(weight) if ctx.needs_input_grad[2] else None
        nb = lib.rsdet_chan_layernorm_ws_size(N, C, H * W)
        ws = torch.empty((nb,), dtype=torch.uint8, device=x.device)
        rc = lib.rsdet_chan_layernorm_backward_f32(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(weight),
                                                   N, C, H * W, _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), nb,
                                                   _lib.stream_ptr())
        _lib.check(rc, "rsdet_chan_layernorm_backward_f32")
        return gx, gw, gb, None


def applies(x, norm):
    return (_ON and type(norm) is torch.nn.LayerNorm and norm.elementwise_affine and norm.bias is not None and x.is_cuda
            and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_autocast_enabled()
            and tuple(norm.normalized_shape) == (x.shape[1],) and norm.weight.dtype == torch.float32
            and 4 <= x.shape[1] <= 512)


def chan_layer_norm(x, norm):
    """``norm(x.flatten(2).transpose(1, 2))`` put back into (B, C, H, W), for an NCHW map that applies() accepted."""
    return _ChanLayerNorm.apply(x, norm.weight, norm.bias, norm.eps)
