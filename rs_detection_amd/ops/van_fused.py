"""The elementwise tails of a VAN block as single passes (csrc/van_ops.hip): the 1x1 convolutions run WITHOUT their bias
and the bias joins the pass that follows it anyway -- GELU, the attention's gate product, or the layer-scale residual.

  bias_gelu(x, b)                    GELU(x + b[c])                       (proj_1 -> activation, van.py:64-66)
  gate(u, a, b)                      u * (a + b[c])                       (LKA: u * conv1(...), van.py:56-60)
  residual(x, p, b, shortcut, ls)    x + ls[c] * (p + b[c] + shortcut)    (Block.execute, van.py:121-122, with the
                                                                           attention's `+ shortcut` of :70 folded in)

Each backward is one pass as well and yields the per-channel gradients (bias, layer scale) from a deterministic
two-stage sum.  fp32 NCHW-contiguous CUDA maps outside autocast; the callers keep the torch expressions for everything
else (CPU, channels_last, bf16) -- same values up to fp32 rounding of a different association
(/root/reference/python/jdet/models/backbones/van.py:46-122)."""
import os

import torch

from .. import _lib

_ON = True      # False: the torch tails (what the fused passes are tested against)


def applies(*maps):
    """All maps: fp32, CUDA, 4-d, NCHW-contiguous, one shape; autocast off; sizes the kernels take."""
    t0 = maps[0]
    if not (_ON and t0.is_cuda and t0.dim() == 4 and not torch.is_autocast_enabled()):
        return False
    for t in maps:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.shape == t0.shape and t.is_contiguous()):
            return False
    N, C, H, W = t0.shape
    return bool(_lib.load().rsdet_van_supported(N, C, H * W))


def _ws(lib, N, C, HW, device):
    nbytes = lib.rsdet_van_ws_size(N, C, HW)
    return torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=device), nbytes


def _vec(t):
    """A per-channel parameter as the fp32 contiguous vector the kernels read (None stays None)."""
    if t is None:
        return None
    return t.detach().float().contiguous() if (t.dtype != torch.float32 or not t.is_contiguous()) else t.detach()


class _BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        b = _vec(bias)
        _lib.check(lib.rsdet_van_bias_gelu_fwd_f32(_lib.ptr(x), _lib.ptr(b), N, C, H * W, _lib.ptr(y), _lib.stream_ptr()),
                   "rsdet_van_bias_gelu_fwd_f32")
        ctx.save_for_backward(x, b)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, b = ctx.saved_tensors
        N, C, H, W = x.shape
        gy = gy.contiguous()
        need_b = ctx.bias_dtype is not None and ctx.needs_input_grad[1]
        gx = torch.empty_like(x)
        gb = torch.empty((C,), dtype=torch.float32, device=x.device) if need_b else None
        ws, nbytes = _ws(lib, N, C, H * W, x.device) if need_b else (None, 0)
        _lib.check(lib.rsdet_van_bias_gelu_bwd_f32(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(b), N, C, H * W, _lib.ptr(gx),
                                                   _lib.ptr(gb), _lib.ptr(ws), nbytes, _lib.stream_ptr()),
                   "rsdet_van_bias_gelu_bwd_f32")
        return gx, (gb.to(ctx.bias_dtype) if need_b else None)


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, a, bias):
        lib = _lib.load()
        N, C, H, W = u.shape
        y = torch.empty_like(u)
        b = _vec(bias)
        _lib.check(lib.rsdet_van_gate_fwd_f32(_lib.ptr(u), _lib.ptr(a), _lib.ptr(b), N, C, H * W, _lib.ptr(y),
                                              _lib.stream_ptr()), "rsdet_van_gate_fwd_f32")
        ctx.save_for_backward(u, a, b)
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        u, a, b = ctx.saved_tensors
        N, C, H, W = u.shape
        g = g.contiguous()
        need_b = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        gu, ga = torch.empty_like(u), torch.empty_like(u)
        gb = torch.empty((C,), dtype=torch.float32, device=u.device) if need_b else None
        ws, nbytes = _ws(lib, N, C, H * W, u.device) if need_b else (None, 0)
        _lib.check(lib.rsdet_van_gate_bwd_f32(_lib.ptr(g), _lib.ptr(u), _lib.ptr(a), _lib.ptr(b), N, C, H * W, _lib.ptr(gu),
                                              _lib.ptr(ga), _lib.ptr(gb), _lib.ptr(ws), nbytes, _lib.stream_ptr()),
                   "rsdet_van_gate_bwd_f32")
        return gu, ga, (gb.to(ctx.bias_dtype) if need_b else None)


class _Residual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, bias, shortcut, scale):
        lib = _lib.load()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        b, ls = _vec(bias), _vec(scale)
        _lib.check(lib.rsdet_van_residual_fwd_f32(_lib.ptr(x), _lib.ptr(p), _lib.ptr(b), _lib.ptr(shortcut), _lib.ptr(ls),
                                                  N, C, H * W, _lib.ptr(y), _lib.stream_ptr()),
                   "rsdet_van_residual_fwd_f32")
        ctx.save_for_backward(p, b, shortcut, ls)
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.scale_dtype = scale.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        p, b, sc, ls = ctx.saved_tensors
        N, C, H, W = p.shape
        g = g.contiguous()
        need_b = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        need_s = ctx.needs_input_grad[4]
        gp = torch.empty_like(p)
        gb = torch.empty((C,), dtype=torch.float32, device=p.device) if need_b else None
        gs = torch.empty((C,), dtype=torch.float32, device=p.device) if need_s else None
        ws, nbytes = _ws(lib, N, C, H * W, p.device)
        _lib.check(lib.rsdet_van_residual_bwd_f32(_lib.ptr(g), _lib.ptr(p), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(ls), N, C,
                                                  H * W, _lib.ptr(gp), _lib.ptr(gb), _lib.ptr(gs), _lib.ptr(ws), nbytes,
                                                  _lib.stream_ptr()), "rsdet_van_residual_bwd_f32")
        return (g if ctx.needs_input_grad[0] else None, gp if ctx.needs_input_grad[1] else None,
                gb.to(ctx.bias_dtype) if need_b else None,
                gp if (sc is not None and ctx.needs_input_grad[3]) else None,
                gs.to(ctx.scale_dtype) if need_s else None)


def bias_gelu(x, bias):
    if applies(x):
        return _BiasGelu.apply(x, bias)
    return torch.nn.functional.gelu(x if bias is None else x + bias.to(x.dtype)[None, :, None, None])


def gate(u, a, bias):
    if applies(u, a):
        return _Gate.apply(u, a, bias)
    return u * (a if bias is None else a + bias.to(a.dtype)[None, :, None, None])


def residual(x, p, bias, shortcut, scale):
    if applies(x, p, shortcut):
        return _Residual.apply(x, p, bias, shortcut, scale)
    f = p if bias is None else p + bias.to(p.dtype)[None, :, None, None]
    if shortcut is not None:
        f = f + shortcut
    return x + scale.to(f.dtype)[None, :, None, None] * f
