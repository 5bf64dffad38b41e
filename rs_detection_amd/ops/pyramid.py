"""The pyramid canvas: the level maps of an FPN batch side by side in ONE tensor, so that the S2ANet head's
shared-weight convolutions run once instead of once per level (csrc/canvas.hip; the reference loops over levels,
/root/reference/python/jdet/models/roi_heads/s2anet_head.py:207-255).

  lay = canvas_layout(sizes, device)                 sizes: [(H0, W0), (H1, W1), ...]
  canvas = pyramid_pack(levels, lay)                 (B,C,Hl,Wl) x L -> (B,C,Hc,Wc), gap pixels zero
  levels = pyramid_unpack(canvas, lay)               the inverse (gaps dropped); each is the other's backward
  y = canvas_bias_act(x, bias, lay, relu)            act(x + bias) with the gap pixels put back to zero

Level 0 sits at the canvas origin, the other levels in a column to its right (``tall=True``: level 1 under it and
the rest beside level 1), one gap pixel between neighbours: the gap is the zero padding a 3x3 / padding-1 convolution of a single level would see, so canvas convolution == per-level
convolution at every level pixel provided the INPUT's gaps are zero -- which pack and canvas_bias_act maintain.
GPU only (no CPU fallback): the callers keep the reference's per-level loop for CPU tensors."""
import ctypes

import numpy as np
import torch

from .. import _lib

__all__ = ["canvas_layout", "pyramid_pack", "pyramid_unpack", "canvas_bias_act", "CanvasLayout"]

GAP = 1


class CanvasLayout:
    """Geometry + the device tables of one pyramid shape: ``rects`` [(y0, x0, H, W)], canvas (Hc, Wc), ``pixmap``
    (int32: -1 gap, else level << 27 | pixel), ``live`` (uint8 per canvas pixel), ``live_f`` ((1,1,Hc,Wc) float)."""

    def __init__(self, sizes, device, align=4, tall=None):
        import os
        sizes = [tuple(int(v) for v in s) for s in sizes]
        assert 1 <= len(sizes) <= 8
        self.sizes = sizes
        tall = bool(tall)
        H0, W0 = sizes[0]
        rects = [(0, 0, H0, W0)]
        if tall and len(sizes) > 1:
            # level 1 under level 0, the remaining levels in a column to the right of level 1: the canvas keeps level
            # 0's width (128 for a 1024^2 tile: what MIOpen's NCHW solvers tile best)
            H1, W1 = sizes[1]
            rects.append((H0 + GAP, 0, H1, W1))
            y = H0 + GAP
            for (h, w) in sizes[2:]:
                rects.append((y, W1 + GAP, h, w))
                y += h + GAP
            Hc = max(H0 + GAP + H1, y - GAP)
            Wc = max(W0, W1 + (GAP + max(w for _, w in sizes[2:]) if len(sizes) > 2 else 0))
        else:
            y = 0
            for (h, w) in sizes[1:]:
                rects.append((y, W0 + GAP, h, w))
                y += h + GAP
            Hc = max(H0, y - GAP if len(sizes) > 1 else 0)
            Wc = W0 + (GAP + max(w for _, w in sizes[1:]) if len(sizes) > 1 else 0)
        Wc = (Wc + align - 1) // align * align          # keeps Hc * Wc a multiple of 4 for the vector kernels
        self.rects, self.Hc, self.Wc = rects, Hc, Wc
        pm = np.full((Hc, Wc), -1, np.int32)
        for l, (y0, x0, h, w) in enumerate(rects):
            assert h * w < (1 << 27)
            pm[y0:y0 + h, x0:x0 + w] = (l << 27) | np.arange(h * w, dtype=np.int32).reshape(h, w)
        self.device = torch.device(device)
        self.pixmap = torch.from_numpy(pm.reshape(-1)).to(self.device)
        self.live = torch.from_numpy((pm.reshape(-1) >= 0).astype(np.uint8)).to(self.device)
        self.live_f = self.live.view(1, 1, Hc, Wc).float()
        self.level_pixels = (ctypes.c_int * len(sizes))(*[h * w for h, w in sizes])

    @property
    def fill(self):
        """Fraction of the canvas that is level pixels."""
        return sum(h * w for h, w in self.sizes) / float(self.Hc * self.Wc)


_LAYOUTS = {}


def canvas_layout(sizes, device):
    key = (tuple(tuple(int(v) for v in s) for s in sizes), str(device))
    if key not in _LAYOUTS:
        if len(_LAYOUTS) > 16:
            _LAYOUTS.clear()
        _LAYOUTS[key] = CanvasLayout(sizes, device)
    return _LAYOUTS[key]


def _is_cl(t):
    """channels-last storage that is not also NCHW-contiguous."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def _copy(levels, canvas, lay, canvas_nhwc, levels_nhwc, to_canvas):
    lib = _lib.load()
    B, C = canvas.shape[:2]
    ptrs = (ctypes.c_void_p * len(levels))(*[t.data_ptr() for t in levels])
    rc = lib.rsdet_pyramid_copy(ptrs, lay.level_pixels, len(levels), _lib.ptr(canvas), _lib.ptr(lay.pixmap), B, C,
                                lay.Hc * lay.Wc, canvas.element_size(), int(canvas_nhwc), int(levels_nhwc),
                                int(to_canvas), _lib.stream_ptr())
    _lib.check(rc, "rsdet_pyramid_copy")


def _as_layout(t, nhwc):
    return t.contiguous(memory_format=torch.channels_last) if nhwc else t.contiguous()


def _empty(shape, dtype, device, nhwc):
    return torch.empty(shape, dtype=dtype, device=device,
                       memory_format=torch.channels_last if nhwc else torch.contiguous_format)


class _Pack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lay, nhwc, *levels):
        lv = [_as_layout(t, nhwc) for t in levels]
        B, C = lv[0].shape[:2]
        canvas = _empty((B, C, lay.Hc, lay.Wc), lv[0].dtype, lv[0].device, nhwc)
        _copy(lv, canvas, lay, nhwc, nhwc, True)
        ctx.lay, ctx.nhwc = lay, nhwc
        return canvas

    @staticmethod
    def backward(ctx, g):
        lay, nhwc = ctx.lay, ctx.nhwc
        g = _as_layout(g, nhwc)
        B, C = g.shape[:2]
        out = [_empty((B, C, h, w), g.dtype, g.device, nhwc) for h, w in lay.sizes]
        _copy(out, g, lay, nhwc, nhwc, False)
        return (None, None) + tuple(out)


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, canvas, lay, out_nhwc):
        nhwc = _is_cl(canvas)
        canvas = _as_layout(canvas, nhwc)
        B, C = canvas.shape[:2]
        out = [_empty((B, C, h, w), canvas.dtype, canvas.device, out_nhwc) for h, w in lay.sizes]
        _copy(out, canvas, lay, nhwc, out_nhwc, False)
        ctx.lay, ctx.nhwc, ctx.out_nhwc, ctx.meta = lay, nhwc, out_nhwc, (B, C, canvas.dtype, canvas.device)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        lay, (B, C, dtype, device) = ctx.lay, ctx.meta
        gl = []
        for g, (h, w) in zip(grads, lay.sizes):
            if g is None:
                g = torch.zeros((B, C, h, w), dtype=dtype, device=device)
            gl.append(_as_layout(g.to(dtype), ctx.out_nhwc))
        canvas = _empty((B, C, lay.Hc, lay.Wc), dtype, device, ctx.nhwc)
        _copy(gl, canvas, lay, ctx.nhwc, ctx.out_nhwc, True)
        return canvas, None, None


def _check(levels, lay):
    t0 = levels[0]
    if not t0.is_cuda:
        raise _lib.RsdetError("rs_detection_amd ops run on the GPU only; no CPU fallback")
    assert len(levels) == len(lay.sizes)
    for t, (h, w) in zip(levels, lay.sizes):
        assert t.dim() == 4 and tuple(t.shape[2:]) == (h, w) and t.shape[:2] == t0.shape[:2] and t.dtype == t0.dtype, \
            (tuple(t.shape), (h, w))
    assert t0.element_size() in (2, 4)


def pyramid_pack(levels, lay, channels_last=None):
    """list of (B,C,Hl,Wl) -> (B,C,Hc,Wc); the canvas takes the levels' memory format (channels_last when the first
    level is) unless ``channels_last`` says otherwise."""
    levels = list(levels)
    _check(levels, lay)
    nhwc = _is_cl(levels[0]) if channels_last is None else bool(channels_last)
    return _Pack.apply(lay, nhwc, *levels)


def pyramid_unpack(canvas, lay, channels_last=False):
    """(B,C,Hc,Wc) -> list of (B,C,Hl,Wl), NCHW-contiguous by default (what the loss kernels and the anchor refinement
    read) whatever the canvas' memory format."""
    if not canvas.is_cuda:
        raise _lib.RsdetError("rs_detection_amd ops run on the GPU only; no CPU fallback")
    assert canvas.dim() == 4 and tuple(canvas.shape[2:]) == (lay.Hc, lay.Wc) and canvas.element_size() in (2, 4)
    return list(_Unpack.apply(canvas, lay, bool(channels_last)))


class _CanvasBiasAct(torch.autograd.Function):
    """Forward: csrc/canvas.hip.  Backward: the bias + activation backward of csrc/bn_act.hip (ops/bn_act._BNAct's
    arithmetic with mean 0, variance 1): for the ReLU form its y > 0 gate already excludes the zeroed gap pixels, for
    the identity form the incoming gradient is masked first."""

    @staticmethod
    def forward(ctx, x, bias, lay, relu):
        lib = _lib.load()
        nhwc = _is_cl(x)
        x = _as_layout(x, nhwc)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        name = "rsdet_canvas_bias_act_" + ("bf16" if x.dtype == torch.bfloat16 else "f32")
        rc = getattr(lib, name)(_lib.ptr(x), _lib.ptr(bias), _lib.ptr(lay.live), N, C, H * W, int(relu), int(nhwc),
                                _lib.ptr(y), _lib.stream_ptr())
        _lib.check(rc, name)
        ctx.save_for_backward(y if relu else None)
        ctx.lay, ctx.relu, ctx.nhwc, ctx.shape = lay, bool(relu), nhwc, (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .bn_act import _UNIT
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        nhwc, lay = ctx.nhwc, ctx.lay
        gy = _as_layout(gy, nhwc)
        if not ctx.relu:
            gy = gy * lay.live_f.to(gy.dtype)
            y = gy                                  # not read by the kernel when relu == 0
        need_x, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = torch.empty_like(gy) if (need_x and ctx.relu) else None
        gb = torch.empty((C,), dtype=torch.float32, device=gy.device) if need_b else None
        ws_size = lib.rsdet_bn_act_backward_nhwc_ws_size if nhwc else lib.rsdet_bn_act_backward_ws_size
        ws_bytes = ws_size(N, C, H * W) if need_b else 0
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gy.device) if ws_bytes else None
        key = (gy.device, C)
        if key not in _UNIT:
            _UNIT[key] = (torch.zeros(C, device=gy.device), torch.ones(C, device=gy.device))
        mean, var = _UNIT[key]
        name = "rsdet_bn_act_backward_" + ("nhwc_" if nhwc else "") + ("bf16" if gy.dtype == torch.bfloat16 else "f32")
        if ctx.relu or need_b:
            rc = getattr(lib, name)(_lib.ptr(gy), _lib.ptr(y), None, _lib.ptr(mean), _lib.ptr(var), None, 0.0, N, C,
                                    H * W, int(ctx.relu), _lib.ptr(gx), None, None, _lib.ptr(gb), _lib.ptr(ws), ws_bytes,
                                    _lib.stream_ptr())
            _lib.check(rc, name)
        return (gx if ctx.relu else gy) if need_x else None, gb, None, None


def canvas_bias_act(x, bias, lay, relu=True):
    """act(x + bias[:, None, None]) on a canvas, gap pixels zero.  Fused for fp32 / bf16 CUDA canvases in NCHW or
    channels_last storage with an fp32 bias; the torch expression otherwise (same result)."""
    if (x.is_cuda and x.dim() == 4 and bias is not None and bias.dtype == torch.float32
            and x.dtype in (torch.float32, torch.bfloat16) and (x.is_contiguous() or (_is_cl(x) and x.shape[1] % 4 == 0))
            and x.shape[0] * x.shape[1] <= 65535 and (x.shape[2] * x.shape[3]) % 4 == 0):
        return _CanvasBiasAct.apply(x, bias, lay, relu)
    out = x if bias is None else x + bias.to(x.dtype)[None, :, None, None]
    if relu:
        out = torch.relu(out)
    return out * lay.live_f.to(out.dtype)
