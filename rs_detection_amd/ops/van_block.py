"""A whole VAN ``Block`` as ONE autograd node (csrc/van_gemm.hip, csrc/dwconv.hip) -- host side.

The reference block (/root/reference/python/jdet/models/backbones/van.py:140-263: Mlp :140-175, AttentionModule :177-192,
SpatialAttention :195-213, Block :216-261) is, on an NCHW fp32 map x,

    xn = norm1(x);  u = GELU(proj_1(xn));  a = conv1(conv_spatial(conv0(u)));  x1 = x + ls1 * (proj_2(u * a) + xn)
    out = x1 + ls2 * fc2(GELU(dwconv(fc1(norm2(x1)))))

with both BatchNorms in TRAINING mode (batch statistics).  Rounds 1-5 ran it as ~40 autograd nodes and ~150 launches
(rocBLAS small-tile GEMMs through MIOpen, NHWC weight-gradient kernels between layout transposes, one elementwise pass per
tail): the Oriented R-CNN / VAN-B3 step was paced by the host (64.6 of 66.4 ms) and carried 70 ms of kernels.  Here:

  * every 1x1 convolution is our streaming fp32 MFMA GEMM on the NCHW map (per image: (O x C) . (C x H W)), and every tail is
    that GEMM's epilogue: bias + GELU (two outputs: t1 and u), bias + gate, bias + layer scale + residual (+ the attention's
    own shortcut), and in the backward the gate's two products, GELU', and the WHOLE BatchNorm backward;
  * a training-mode BatchNorm is never applied to a map: its statistics are one reduction pass, its affine map
    xn = x sc + sh is folded into the weights and bias of the convolution behind it (W sc[k], b + W sh), the shortcut's xn
    into the residual epilogue's constants; in the backward its sums come from the weight gradient (sum_p gxn x =
    sum_o W[o,k] UT[k,o], sum_p gxn = sum_o W[o,k] gs[o]) and its map gx = sc (gxn - c1 - xhat c2) is the epilogue of the
    backward-data GEMM -- no pass over a map belongs to a BatchNorm;
  * layer scales ride in transposed weights (backward-data) and in the weight-gradient folds; their own gradients are the
    folds' row dots (the scaled branch output is never stored);
  * weight gradients are split-K MFMA GEMMs on the NCHW maps as they lie (both operands pixel-contiguous): no NHWC round
    trip, no transposes, no zero fills.

13 launches forward, 25 backward -- issued by TWO C calls (csrc/van_block.hip: rsdet_van_block_forward_f32 /
_backward_f32) into three arenas per block (saved activations, scratch, gradients), so the host side of a block is two
ctypes calls and five allocations instead of ~60 of each.  Everything is exact fp32 (the
GEMMs are k-ordered fmaf chains); tests/test_gpu_van_block.py pins output and all 23 gradients against fp32 autograd of the
per-operator composite (<= 1e-4 relative).

Applies to: CUDA fp32 NCHW-contiguous x outside autocast, BatchNorms in training mode with running statistics, idle
drop-path / dropout, exact GELU, channel counts the GEMM tiles divide (every stage of VAN-B1..B3 at image sizes whose
H W is a multiple of 128).  Anything else runs the per-operator forward of models/backbones/van.py."""
import ctypes

import torch

from .. import _lib

_ON = True      # False: the per-operator route (what this node is tested against)

_P5, _I5 = ctypes.c_void_p * 5, ctypes.c_int * 5


def _p(t):
    return t.data_ptr() if t is not None else None


def _desc(x, params, bn1, bn2):
    """struct rsdet_van_block for this call (shapes, parameter pointers, the BatchNorms' running statistics)."""
    N, C, H, W = x.shape
    d = _lib.VanBlock()
    d.N, d.C, d.H, d.W, d.R = N, C, H, W, params[15].shape[0]            # (params[15] = fc1's weight)
    for name, t in zip(_lib.VanBlock.PARAMS, params):
        setattr(d, name, t.data_ptr())
    (d.rm1, d.rv1, d.nbt1, d.eps1, d.mom1), (d.rm2, d.rv2, d.nbt2, d.eps2, d.mom2) = \
        [(rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), eps, mom) for rm, rv, nbt, eps, mom in (bn1, bn2)]
    return d


_SIZES = {}     # (N, C, H, W, R) -> (saved, forward scratch, backward scratch, gradients) in floats


def _sizes(lib, d):
    key = (d.N, d.C, d.H, d.W, d.R)
    v = _SIZES.get(key)
    if v is None:
        r = ctypes.byref(d)
        v = _SIZES[key] = (lib.rsdet_van_block_saved_floats(r), lib.rsdet_van_block_forward_scratch_floats(r),
                           lib.rsdet_van_block_backward_scratch_floats(r), lib.rsdet_van_block_grad_floats(r))
    return v


def _up4(n):
    return (n + 3) & ~3


class _VanBlock(torch.autograd.Function):
    """forward(x, *22 parameters, bn1, bn2): the parameters in the order of _lib.VanBlock.PARAMS."""

    @staticmethod
    def forward(ctx, x, *args):
        lib = _lib.load()
        params, bn1, bn2 = args[:22], args[22], args[23]
        d = _desc(x, params, bn1, bn2)
        n_saved, n_fwd, _, _ = _sizes(lib, d)
        saved = torch.empty((n_saved,), dtype=torch.float32, device=x.device)
        scratch = torch.empty((n_fwd,), dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        rc = lib.rsdet_van_block_forward_f32(ctypes.byref(d), x.data_ptr(), out.data_ptr(), saved.data_ptr(),
                                             scratch.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "rsdet_van_block_forward_f32")
        ctx.save_for_backward(x, saved, *params)
        ctx.bn = (bn1, bn2)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        x, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        d = _desc(x, params, *ctx.bn)
        _, _, n_bwd, n_grad = _sizes(lib, d)
        gout = gout.contiguous()
        scratch = torch.empty((n_bwd,), dtype=torch.float32, device=x.device)
        grads = torch.empty((n_grad,), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        rc = lib.rsdet_van_block_backward_f32(ctypes.byref(d), x.data_ptr(), gout.data_ptr(), saved.data_ptr(),
                                              scratch.data_ptr(), _p(gx), grads.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "rsdet_van_block_backward_f32")
        # the flat gradient arena -> one view per parameter (16-byte aligned slices in parameter order)
        out, off = [gx], 0
        for p in params:
            n = p.numel()
            out.append(grads[off:off + n].view(p.shape))
            off += _up4(n)
        out += [None, None]
        return tuple(out)


def _conv_ok(conv, k, dil=1, groups=1):
    return (conv.kernel_size == (k, k) and conv.stride == (1, 1) and conv.dilation == (dil, dil) and conv.groups == groups
            and conv.bias is not None and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous()
            and conv.padding == (dil * (k - 1) // 2,) * 2 and conv.padding_mode == "zeros")


def _bn_ok(bn):
    return (type(bn) is torch.nn.BatchNorm2d and bn.training and bn.track_running_stats and bn.affine
            and bn.momentum is not None and bn.weight.dtype == torch.float32)


def _static(block):
    """The checks that depend on the module's construction alone, once per block: (ok, parameters in _lib.VanBlock.PARAMS
    order, the two BatchNorms)."""
    st = block.__dict__.get("_vb_static")
    if st is None:
        at, mlp = block.attn, block.mlp
        sg = at.spatial_gating_unit
        C, R = at.proj_1.in_channels, mlp.fc1.out_channels
        ok = (isinstance(at.activation, torch.nn.GELU) and at.activation.approximate == "none"
              and isinstance(mlp.act, torch.nn.GELU) and mlp.act.approximate == "none"
              and _conv_ok(at.proj_1, 1) and _conv_ok(sg.conv1, 1) and _conv_ok(at.proj_2, 1) and _conv_ok(mlp.fc1, 1)
              and _conv_ok(mlp.fc2, 1) and _conv_ok(sg.conv0, 5, 1, C) and _conv_ok(sg.conv_spatial, 7, 3, C)
              and _conv_ok(mlp.dwconv.dwconv, 3, 1, R))
        n1, n2, dw = block.norm1, block.norm2, mlp.dwconv.dwconv
        params = (n1.weight, n1.bias, at.proj_1.weight, at.proj_1.bias, sg.conv0.weight, sg.conv0.bias, sg.conv_spatial.weight,
                  sg.conv_spatial.bias, sg.conv1.weight, sg.conv1.bias, at.proj_2.weight, at.proj_2.bias, block.layer_scale_1,
                  n2.weight, n2.bias, mlp.fc1.weight, mlp.fc1.bias, dw.weight, dw.bias, mlp.fc2.weight, mlp.fc2.bias,
                  block.layer_scale_2)
        st = block.__dict__["_vb_static"] = (ok, params, R)
    return st


def applies(block, x):
    """Does the one-node form take this models/backbones/van.py Block on this input?  (module docstring)"""
    if not (_ON and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
            and not torch.is_autocast_enabled() and torch.is_grad_enabled()):
        return False
    ok, params, R = _static(block)
    if not (ok and _bn_ok(block.norm1) and _bn_ok(block.norm2)):
        return False
    for p in params:
        if not (p.requires_grad and p.dtype == torch.float32):
            return False
    N, C, H, W = x.shape
    return _shape_ok(N, C, H, W, R)


_SHAPE_OK = {}


def _shape_ok(N, C, H, W, R):
    key = (N, C, H, W, R)
    v = _SHAPE_OK.get(key)
    if v is None:
        d = _lib.VanBlock()
        d.N, d.C, d.H, d.W, d.R = key
        v = _SHAPE_OK[key] = bool(_lib.load().rsdet_van_block_supported(ctypes.byref(d)))
    return v


def van_block(block, x):
    """``block(x)`` for a Block that applies() accepted."""
    n1, n2 = block.norm1, block.norm2
    bn1 = (n1.running_mean, n1.running_var, n1.num_batches_tracked, float(n1.eps), float(n1.momentum))
    bn2 = (n2.running_mean, n2.running_var, n2.num_batches_tracked, float(n2.eps), float(n2.momentum))
    return _VanBlock.apply(x, *_static(block)[1], bn1, bn2)
