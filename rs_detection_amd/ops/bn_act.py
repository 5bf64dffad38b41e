"""Eval-mode BatchNorm + residual add + ReLU as ONE HBM pass each way (csrc/bn_act.hip).

The backbone keeps every BatchNorm in eval mode (resnet.py:177-184, norm_eval), so ``bn -> (+ identity) -> relu``
(resnet.py:101-126) is a per-channel affine followed by two elementwise ops: three kernels forward and three
backward on the torch route, ~8 ms of a 68 ms S2ANet step.  ``bn_act(x, bn, residual, relu)`` fuses them when it
can (CUDA, contiguous NCHW, fp32 activations or the bf16 ones of an autocast step, ``bn`` in eval mode) and is the
plain torch sequence otherwise (training-mode BatchNorm, CPU tensors of the RetinaNet plumbing case) -- both are the product path; there is no
oracle or CPU restatement behind it."""
import os

import torch
import torch.nn.functional as F

from rs_detection_amd import _lib


_RELU_MASK = True   # the backward reads the one-bit ReLU gate (False: it reads y; kept for the equivalence test)


_SHAPE_MEMO = {}     # shape-derived answers of the library (mask bytes, workspace bytes, "NHWC kernels take C"): one ctypes
                     # call each per (question, shape) instead of per invocation -- the trunk asks ~150 times per step


def _memo(fn_name, *args):
    key = (fn_name,) + args
    v = _SHAPE_MEMO.get(key)
    if v is None:
        if len(_SHAPE_MEMO) > 4096:
            _SHAPE_MEMO.clear()
        v = _SHAPE_MEMO[key] = getattr(_lib.load(), fn_name)(*args)
    return v


class Forked:
    """The output of a residual block handed on as TWO autograd outputs of the node that produced it (same storage): `a` for
    the next block's first convolution, `b` for its identity branch.  The two gradients then reach the producing node
    separately and its backward kernel sums them while reading (rsdet_bn_act_backward_nhwc_mask2_f32) -- autograd's own
    accumulation is a pass of its own over the trunk's widest tensors (0.5 ms of the fp32 S2ANet step)."""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b


_FORK = os.environ.get("RSDET_BN_FORK", "1") != "0"     # off: one output, autograd adds the gradients (the form it is tested against)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, mean, var, eps, relu, fork=False):
        lib = _lib.load()
        ctx.fork = bool(fork)
        N, C, H, W = x.shape
        # channels_last tensors (the bf16 trunk) go to the NHWC kernels; the output keeps the input's layout
        ctx.nhwc = not x.is_contiguous()
        y = torch.empty_like(x)
        tag = "bf16" if x.dtype == torch.bfloat16 else "f32"
        # channels_last + ReLU + a backward to come: the forward leaves the ReLU gate as one bit per element and the
        # backward reads that instead of y (19 % less traffic there)
        mask = None
        if ctx.nhwc and relu and _RELU_MASK and any(ctx.needs_input_grad[:4]):
            nb = _memo("rsdet_bn_act_relu_mask_bytes", N, C, H * W, int(tag == "bf16"))
            if nb:
                mask = torch.empty((nb,), dtype=torch.uint8, device=x.device)
        if mask is not None:
            name = "rsdet_bn_act_forward_nhwc_mask_" + tag
            rc = getattr(lib, name)(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(weight),
                                    _lib.ptr(bias), float(eps), N, C, H * W, 1, _lib.ptr(y), _lib.ptr(mask),
                                    _lib.stream_ptr())
        else:
            name = "rsdet_bn_act_forward_" + ("nhwc_" if ctx.nhwc else "") + tag
            rc = getattr(lib, name)(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(weight),
                                    _lib.ptr(bias), float(eps), N, C, H * W, int(relu), _lib.ptr(y), _lib.stream_ptr())
        _lib.check(rc, name)
        # x is only needed for the weight gradient (sum of g * xhat); y only for the ReLU gate when there is no mask
        ctx.has_mask = mask is not None
        ctx.save_for_backward(x if weight is not None else None, mask if mask is not None else y, weight, mean, var)
        ctx.y_dtype, ctx.y_device = y.dtype, y.device
        ctx.shape = tuple(x.shape)
        ctx.eps, ctx.relu, ctx.has_res = float(eps), bool(relu), residual is not None
        ctx.has_bias = bias is not None
        if ctx.fork:
            return y, y.view_as(y)
        return y

    @staticmethod
    def backward(ctx, gy, gy2=None):
        lib = _lib.load()
        x, y, weight, mean, var = ctx.saved_tensors          # (y is the bit mask when ctx.has_mask)
        N, C, H, W = ctx.shape
        fmt = torch.channels_last if ctx.nhwc else torch.contiguous_format
        if gy is None:
            gy, gy2 = gy2, None
        if gy2 is not None:
            gy2 = gy2.contiguous(memory_format=fmt).to(ctx.y_dtype)
            if not (ctx.has_mask and ctx.y_dtype == torch.float32):
                gy, gy2 = gy + gy2, None             # (the kernels of the other layouts / dtypes take one gradient)
        gy = gy.contiguous(memory_format=fmt).to(ctx.y_dtype)
        need_x, need_res = ctx.needs_input_grad[0], ctx.has_res and ctx.needs_input_grad[1]
        need_w = weight is not None and ctx.needs_input_grad[2]
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        gx = torch.empty_like(gy) if need_x else None
        # without a ReLU the residual's gradient IS grad_y: no copy
        gres = (torch.empty_like(gy) if ctx.relu else gy) if need_res else None
        gw = torch.empty_like(weight) if need_w else None
        gb = torch.empty_like(mean) if need_b else None
        ws_bytes = _memo("rsdet_bn_act_backward_nhwc_ws_size" if ctx.nhwc else "rsdet_bn_act_backward_ws_size", N, C,
                         H * W) if (need_w or need_b) else 0
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gy.device) if ws_bytes else None
        tag = "bf16" if ctx.y_dtype == torch.bfloat16 else "f32"
        if ctx.has_mask and gy2 is not None:
            name = "rsdet_bn_act_backward_nhwc_mask2_f32"
            rc = lib.rsdet_bn_act_backward_nhwc_mask2_f32(_lib.ptr(gy), _lib.ptr(gy2), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean),
                                                          _lib.ptr(var), _lib.ptr(weight), ctx.eps, N, C, H * W, _lib.ptr(gx),
                                                          _lib.ptr(gres) if need_res else None, _lib.ptr(gw), _lib.ptr(gb),
                                                          _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        elif ctx.has_mask:
            name = "rsdet_bn_act_backward_nhwc_mask_" + tag
            rc = getattr(lib, name)(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(var),
                                    _lib.ptr(weight), ctx.eps, N, C, H * W, _lib.ptr(gx),
                                    _lib.ptr(gres) if need_res else None, _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws),
                                    ws_bytes, _lib.stream_ptr())
        else:
            name = "rsdet_bn_act_backward_" + ("nhwc_" if ctx.nhwc else "") + tag
            rc = getattr(lib, name)(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(var),
                                    _lib.ptr(weight), ctx.eps, N, C, H * W, int(ctx.relu), _lib.ptr(gx),
                                    _lib.ptr(gres) if (need_res and ctx.relu) else None, _lib.ptr(gw),
                                    _lib.ptr(gb), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        _lib.check(rc, name)
        return gx, gres, gw, gb, None, None, None, None, None


_NO_FUSED_BN = False   # True: torch's batch_norm + relu (what the fused kernels are tested against)


def _layout_ok(x, other=None):
    """NCHW-contiguous, or channels_last-contiguous with a channel count the NHWC kernels take; ``other`` (residual)
    must share the layout."""
    if x.is_contiguous():
        return other is None or other.is_contiguous()
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        if other is not None and not (other.is_contiguous(memory_format=torch.channels_last) and not other.is_contiguous()):
            return False
        return bool(_memo("rsdet_bn_act_nhwc_supported", int(x.shape[1])))
    return False


def _fusable(x, bn, residual):
    if _NO_FUSED_BN:
        return False
    # fp32 activations outside autocast, or bf16 activations (what the convolutions emit under bf16 autocast); the
    # BatchNorm parameters and running statistics are fp32 in both cases
    ok_dtype = (x.dtype == torch.float32 and not torch.is_autocast_enabled()) or x.dtype == torch.bfloat16
    return (x.is_cuda and ok_dtype and x.dim() == 4 and not bn.training
            and bn.running_mean is not None and bn.running_mean.dtype == torch.float32
            and (bn.weight is None or bn.weight.dtype == torch.float32)
            and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape))
            and _layout_ok(x, residual))


def bn_act(x, bn, residual=None, relu=True, fork=False):
    """relu(bn(x) + residual) with ``bn`` an ``nn.BatchNorm2d``.  ``fork``: the result as a Forked pair (see there) when the
    fused node takes it and a gradient will flow; a plain tensor otherwise."""
    if _fusable(x, bn, residual):
        if fork and _FORK and relu and torch.is_grad_enabled() and x.requires_grad and x.dtype == torch.float32 \
                and not x.is_contiguous():
            return Forked(*_BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu, True))
        return _BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


_UNIT = {}


def scale_residual(x, f, scale):
    """x + scale[:, None, None] * f  (layer scale + residual of the VAN block, van.py:121-122 of the reference) through
    the same fused kernels: a per-channel affine with mean 0, variance 1, eps 0 and no bias, plus the residual -- one
    pass forward, one backward (grad_f, and the deterministic two-stage sum for grad_scale) instead of two and four
    torch kernels.  Falls back to the torch expression where the fused path does not apply."""
    if (x.is_cuda and x.dtype == torch.float32 and f.dtype == torch.float32 and x.dim() == 4 and x.shape == f.shape
            and x.is_contiguous() and f.is_contiguous() and scale.dtype == torch.float32
            and not torch.is_autocast_enabled()):
        key = (x.device, x.shape[1])
        if key not in _UNIT:
            _UNIT[key] = (torch.zeros(x.shape[1], device=x.device), torch.ones(x.shape[1], device=x.device))
        mean, var = _UNIT[key]
        return _BNAct.apply(f, x, scale, None, mean, var, 0.0, False)
    return x + scale[:, None, None] * f


class _BiasAddCL(torch.autograd.Function):
    """y = x + bias for a channels_last map with few channels (the prediction maps of the S2ANet head): torch's add
    forward, and a bias gradient from rsdet_colsum_* instead of torch's strided reduction (~20 us per call there)."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.bias_dtype = bias.dtype
        return x + bias.to(x.dtype).view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        gb = None
        if ctx.needs_input_grad[1]:
            N, C, H, W = gy.shape
            g = gy.permute(0, 2, 3, 1)
            if g.is_contiguous() and C <= 64 and gy.dtype in (torch.float32, torch.bfloat16):
                lib = _lib.load()
                rows = N * H * W
                gb32 = torch.empty((C,), dtype=torch.float32, device=gy.device)
                ws_bytes = lib.rsdet_colsum_ws_size(rows, C)
                ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=gy.device)
                name = "rsdet_colsum_bf16" if gy.dtype == torch.bfloat16 else "rsdet_colsum_f32"
                _lib.check(getattr(lib, name)(_lib.ptr(g), rows, C, _lib.ptr(gb32), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr()), name)
                gb = gb32.to(ctx.bias_dtype)
            else:
                gb = gy.sum((0, 2, 3), dtype=torch.float32).to(ctx.bias_dtype)
        return gy, gb


def conv2d_bias(conv, x):
    """``conv(x)`` for a plain nn.Conv2d with a bias and no activation.  A channels_last input under bf16 autocast gets
    the convolution without its bias and the bias as a separate pass whose backward is OUR reduction: the fused
    NHWC bias pass (``bias_act(relu=False)``) for the wide maps, torch's add + rsdet_colsum for maps with few channels.
    Everything else: the module itself."""
    fp32_few = (not torch.is_autocast_enabled() and x.dtype == torch.float32 and conv.out_channels <= 64
                and conv.bias is not None and conv.bias.dtype == torch.float32)   # the fp32 step in channels_last
    if (type(conv) is torch.nn.Conv2d and conv.bias is not None and conv.padding_mode == 'zeros' and x.is_cuda
            and x.dim() == 4 and (torch.is_autocast_enabled() or fp32_few) and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and not _NO_FUSED_BN):
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if y.is_contiguous(memory_format=torch.channels_last) or y.shape[1] == 1:
            C = y.shape[1]
            if _layout_ok(y) and not y.is_contiguous():
                return bias_act(y, conv.bias, relu=False)
            if C <= 64:
                return _BiasAddCL.apply(y, conv.bias)
        return y + conv.bias.to(y.dtype).view(1, -1, 1, 1)
    return conv(x)


def bias_act(x, bias, relu=True):
    """relu(x + bias[:, None, None]): the epilogue of a convolution launched without its bias (ConvModule of the
    detection heads).  One pass forward, one backward that also yields the bias gradient (deterministic two-stage
    sum) -- instead of MIOpen's bias kernel + clamp forward and threshold + reduction backward."""
    if (x.is_cuda and x.dim() == 4 and bias is not None and bias.dtype == torch.float32 and _layout_ok(x)
            and ((x.dtype == torch.float32 and not torch.is_autocast_enabled()) or x.dtype == torch.bfloat16)):
        key = (x.device, x.shape[1])
        if key not in _UNIT:
            _UNIT[key] = (torch.zeros(x.shape[1], device=x.device), torch.ones(x.shape[1], device=x.device))
        mean, var = _UNIT[key]
        return _BNAct.apply(x, None, None, bias, mean, var, 0.0, relu)
    out = x + bias.to(x.dtype)[None, :, None, None]
    return F.relu(out) if relu else out


def bn_relu_maxpool(x, bn, pool):
    """``pool(relu(bn(x)))`` -- the ResNet stem tail (models/backbones/resnet.py:186-189 of the reference).  One fused
    forward pass (csrc/bn_act.hip: bn_relu_maxpool_nhwc_kernel) when no gradient is recorded (the stem is frozen in every
    shipped config, or inference), the map is channels_last on the GPU, ``bn`` is in eval mode and ``pool`` is the
    3x3 / stride 2 / padding 1 max-pool: the full-resolution activation is never written.  The unfused sequence otherwise."""
    C = x.shape[1] if x.dim() == 4 else 0
    if (not torch.is_grad_enabled() and not _NO_FUSED_BN and x.is_cuda and x.dim() == 4 and not bn.training
            and x.dtype in (torch.float32, torch.bfloat16) and C % (8 if x.dtype == torch.bfloat16 else 4) == 0
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and bn.running_mean is not None and bn.running_mean.dtype == torch.float32
            and (bn.weight is None or bn.weight.dtype == torch.float32)
            and isinstance(pool, torch.nn.MaxPool2d) and _pair2(pool.kernel_size) == (3, 3) and _pair2(pool.stride) == (2, 2)
            and _pair2(pool.padding) == (1, 1) and _pair2(pool.dilation) == (1, 1) and not pool.ceil_mode
            and not pool.return_indices
            and (x.dtype == torch.bfloat16 or not torch.is_autocast_enabled())):
        lib = _lib.load()
        N, _, H, W = x.shape
        y = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last)
        rc = lib.rsdet_bn_relu_maxpool_nhwc(_lib.ptr(x), int(x.dtype == torch.bfloat16), _lib.ptr(bn.running_mean),
                                            _lib.ptr(bn.running_var), _lib.ptr(bn.weight), _lib.ptr(bn.bias), float(bn.eps),
                                            N, C, H, W, _lib.ptr(y), _lib.stream_ptr())
        _lib.check(rc, "rsdet_bn_relu_maxpool_nhwc")
        return y
    return pool(bn_act(x, bn))


def _pair2(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
