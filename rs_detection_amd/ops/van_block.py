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

13 launches forward, 29 backward, two Python-level nodes' worth of host work per block.  Everything is exact fp32 (the
GEMMs are k-ordered fmaf chains); tests/test_gpu_van_block.py pins output and all 23 gradients against fp32 autograd of the
per-operator composite (<= 1e-4 relative).

Applies to: CUDA fp32 NCHW-contiguous x outside autocast, BatchNorms in training mode with running statistics, idle
drop-path / dropout, exact GELU, channel counts the GEMM tiles divide (every stage of VAN-B1..B3 at image sizes whose
H W is a multiple of 128).  Anything else runs the per-operator forward of models/backbones/van.py."""
import ctypes

import torch

from .. import _lib

_ON = True      # False: the per-operator route (what this node is tested against)

_E_NONE, _E_BIAS, _E_BIAS_GELU2, _E_GATE2, _E_AFFINE, _E_MUL2, _E_MUL1 = range(7)
_P5, _I5 = ctypes.c_void_p * 5, ctypes.c_int * 5


def _p(t):
    return t.data_ptr() if t is not None else None


def _gemm(lib, st, w, x, M, K, P, n, epi, out0, out1=None, v=(None, None, None, None), s=(None, None)):
    rc = lib.rsdet_van_gemm_f32(_p(w), _p(x), M, K, P, n, epi, _p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]), _p(s[0]), _p(s[1]),
                                _p(out0), _p(out1), st)
    _lib.check(rc, "rsdet_van_gemm_f32")


def _reduce(lib, st, a, b, N, C, P, mode):
    ns = N * lib.rsdet_van_chan_slices(P)
    tab = torch.empty((C, ns, 2), dtype=torch.float32, device=a.device)
    _lib.check(lib.rsdet_van_chan_reduce_f32(_p(a), _p(b), N, C, P, mode, _p(tab), st), "rsdet_van_chan_reduce_f32")
    return tab, ns


def _wgrad(lib, st, g, x, M, Nn, P, n):
    S = lib.rsdet_van_wgrad_f32_splits(M, Nn, P, n)
    part = torch.empty((S, M, Nn), dtype=torch.float32, device=g.device)
    _lib.check(lib.rsdet_van_wgrad_f32(_p(g), _p(x), M, Nn, P, n, _p(part), st), "rsdet_van_wgrad_f32")
    return part, S


def _dw_fwd(lib, st, x, w, b, N, C, H, W, K, dil, act=False):
    y = torch.empty_like(x)
    if act:
        y2 = torch.empty_like(x)
        # (y = GELU'(conv): all the activation's backward needs; y2 = GELU(conv))
        _lib.check(lib.rsdet_dwconv2d_forward_act_f32(_p(x), _p(w), _p(b), N, C, H, W, K, dil, 1, _p(y), _p(y2), st),
                   "rsdet_dwconv2d_forward_act_f32")
        return y, y2
    _lib.check(lib.rsdet_dwconv2d_forward_f32(_p(x), None, _p(w), _p(b), N, C, H, W, K, dil, _p(y), st),
               "rsdet_dwconv2d_forward_f32")
    return y


def _dw_bwd_w(lib, st, gy, x, w, N, C, H, W, K, dil):
    gw = torch.empty_like(w)
    gb = torch.empty((C,), dtype=torch.float32, device=gy.device)
    nb = lib.rsdet_dwconv2d_backward_weight_ws_size(N, C, H, W, K)
    ws = torch.empty((max(nb, 4),), dtype=torch.uint8, device=gy.device)
    _lib.check(lib.rsdet_dwconv2d_backward_weight_f32(_p(gy), _p(x), None, N, C, H, W, K, dil, _p(gw), _p(gb), _p(ws), nb, st),
               "rsdet_dwconv2d_backward_weight_f32")
    return gw, gb


class _VanBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g1, be1, wp1, bp1, wd5, bd5, wd7, bd7, wc1, bc1, wp2, bp2, ls1, g2, be2, wf1, bf1, wd3, bd3, wf2,
                bf2, ls2, bn1, bn2):
        lib = _lib.load()
        st = _lib.stream_ptr()
        N, C, H, W = x.shape
        P, R = H * W, wf1.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        sl = lib.rsdet_van_chan_slices(P)
        ns, ln = N * sl, P // sl

        def bn_fold(inp, gamma, beta, w, b, O, bn, ls, b2, shortcut):
            """statistics of inp -> (w sc, b + w sh), the saved (mean, rstd, sc, sh), the residual epilogue's constants"""
            tab, _ = _reduce(lib, st, inp, None, N, C, P, 1)
            wf, bf_ = torch.empty((O, C), **f32), torch.empty((O,), **f32)
            stats, e = torch.empty((4, C), **f32), torch.empty((2, C), **f32)
            rm, rv, nbt, eps, mom = bn
            f = _lib.VanBnPrep(_p(tab), _p(gamma), _p(beta), _p(w), _p(b), _p(wf), _p(bf_), stats[0].data_ptr(),
                               stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), _p(rm), _p(rv), _p(nbt),
                               _p(ls), _p(b2), e[0].data_ptr(), e[1].data_ptr(), int(shortcut), O, C, ns, ln, eps, mom)
            _lib.check(lib.rsdet_van_bn_prep_f32(ctypes.byref(f), st), "rsdet_van_bn_prep_f32")
            return wf, bf_, stats, e
        # ---- attention half
        w1f, b1f, st1, e1 = bn_fold(x, g1, be1, wp1, bp1, C, bn1, ls1, bp2, True)
        t1, u = torch.empty_like(x), torch.empty_like(x)
        _gemm(lib, st, w1f, x, C, C, P, N, _E_BIAS_GELU2, t1, u, v=(b1f, None, None, None))
        a0 = _dw_fwd(lib, st, u, wd5, bd5, N, C, H, W, 5, 1)
        a1 = _dw_fwd(lib, st, a0, wd7, bd7, N, C, H, W, 7, 3)
        a2, gt = torch.empty_like(x), torch.empty_like(x)
        _gemm(lib, st, wc1, a1, C, C, P, N, _E_GATE2, a2, gt, v=(bc1, None, None, None), s=(u, None))
        x1 = torch.empty_like(x)
        _gemm(lib, st, wp2, gt, C, C, P, N, _E_AFFINE, x1, v=(e1[0], ls1, e1[1], None), s=(x, None))
        # ---- MLP half
        w4f, b4f, st2, e2 = bn_fold(x1, g2, be2, wf1, bf1, R, bn2, ls2, bf2, False)
        h = torch.empty((N, R, H, W), **f32)
        _gemm(lib, st, w4f, x1, R, C, P, N, _E_BIAS, h, v=(b4f, None, None, None))
        d3, h3 = _dw_fwd(lib, st, h, wd3, bd3, N, R, H, W, 3, 1, act=True)      # GELU'(dwconv(h)), GELU(dwconv(h))
        out = torch.empty_like(x)
        _gemm(lib, st, wf2, h3, C, R, P, N, _E_AFFINE, out, v=(None, ls2, e2[1], None), s=(x1, None))
        # ---- the backward-data operands (functions of the parameters alone): five transposes, one launch
        wt = [torch.empty((C, C), **f32), torch.empty((C, C), **f32), torch.empty((C, C), **f32), torch.empty((C, R), **f32),
              torch.empty((R, C), **f32)]
        rc = lib.rsdet_van_transposes_f32(5, _P5(_p(wp1), _p(wc1), _p(wp2), _p(wf1), _p(wf2)),
                                          _P5(None, None, _p(ls1), None, _p(ls2)), _P5(*[_p(t) for t in wt]),
                                          _I5(C, C, C, R, C), _I5(C, C, C, C, R), st)
        _lib.check(rc, "rsdet_van_transposes_f32")
        ctx.save_for_backward(x, t1, u, a0, a1, a2, gt, x1, h, d3, h3, st1, st2, wp1, wd5, wd7, wp2, bp2, ls1, wf1, wd3,
                              wf2, bf2, ls2, *wt)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        st = _lib.stream_ptr()
        (x, t1, u, a0, a1, a2, gt, x1, h, d3, h3, st1, st2, wp1, wd5, wd7, wp2, bp2, ls1, wf1, wd3, wf2, bf2, ls2,
         w1t, w2t, w3t, w4t, w5t) = ctx.saved_tensors
        N, C, H, W = x.shape
        P, R = H * W, wf1.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        gout = gout.contiguous()
        cnt = float(N * P)

        def fold_rows(part, S, M, Nn, rs, w, gs_tab, gs_ns, gs_stride, bias, r_tab, r_ns, sc, sh, want_rs):
            gw = torch.empty((M, Nn), **f32)
            gb = torch.empty((M,), **f32)
            grs = torch.empty((M,), **f32) if want_rs else None
            f = _lib.VanRowsFold(_p(part), _p(rs), _p(w), _p(gs_tab), _p(bias), _p(r_tab), _p(sc), _p(sh), _p(gw), _p(gb),
                                 _p(grs), S, M, Nn, gs_ns, gs_stride, r_ns)
            _lib.check(lib.rsdet_van_fold_rows_f32(ctypes.byref(f), st), "rsdet_van_fold_rows_f32")
            return gw, gb, grs

        def fold_bn(part, S, K, O, wt, gs, r_tab, r_ns, ls, stats):
            gw, gb = torch.empty((O, K), **f32), torch.empty((O,), **f32)
            vec = torch.empty((6, K), **f32)                      # grad_gamma, grad_beta, v0..v3
            f = _lib.VanBnFold(_p(part), _p(wt), _p(gs), _p(r_tab), _p(ls), stats[0].data_ptr(), stats[1].data_ptr(),
                               stats[2].data_ptr(), stats[3].data_ptr(), _p(gw), _p(gb), vec[0].data_ptr(),
                               vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(),
                               vec[5].data_ptr(), S, K, O, 1, 1, r_ns, cnt)
            _lib.check(lib.rsdet_van_fold_bn_f32(ctypes.byref(f), st), "rsdet_van_fold_bn_f32")
            return gw, gb, vec
        # ================= MLP half: out = x1 + ls2 (fc2(h3) + bf2)
        tabg, nsg = _reduce(lib, st, gout, None, N, C, P, 0)                      # sum_p gout
        part, S = _wgrad(lib, st, gout, h3, C, R, P, N)
        gwf2, gbf2, gls2 = fold_rows(part, S, C, R, ls2, wf2, tabg, nsg, 2, bf2, None, 0, None, None, True)
        gh2 = torch.empty_like(d3)                                                # through fc2 and the GELU
        _gemm(lib, st, w5t, gout, R, C, P, N, _E_MUL1, gh2, s=(d3, None))
        gh, gsh = torch.empty_like(h), torch.empty((R,), **f32)                   # through the depthwise 3x3
        nb = lib.rsdet_dwconv2d_backward_data_ws_size(N, R, H, W)
        ws = torch.empty((max(nb, 4),), dtype=torch.uint8, device=dev)
        _lib.check(lib.rsdet_dwconv2d_backward_data_f32(_p(gh2), _p(wd3), N, R, H, W, 3, 1, _p(gh), _p(gsh), _p(ws), nb, st),
                   "rsdet_dwconv2d_backward_data_f32")
        gwd3, gbd3 = _dw_bwd_w(lib, st, gh2, h, wd3, N, R, H, W, 3, 1)
        del gh2
        part, S = _wgrad(lib, st, x1, gh, C, R, P, N)                             # UT (C, R) against the RAW x1
        gwf1, gbf1, vec2 = fold_bn(part, S, C, R, w4t, gsh, None, 0, None, st2)
        G = torch.empty_like(x)                                                   # the gradient of x1, norm2's backward included
        _gemm(lib, st, w4t, gh, C, R, P, N, _E_AFFINE, G, v=(vec2[2], vec2[3], vec2[4], vec2[5]), s=(gout, x1))
        del gh
        # ================= attention half: x1 = x + ls1 (proj_2(gt) + bp2 + xn)
        tabr, nsr = _reduce(lib, st, G, x, N, C, P, 0)                            # (sum_p G, sum_p G x)
        part, S = _wgrad(lib, st, G, gt, C, C, P, N)
        gwp2, gbp2, gls1 = fold_rows(part, S, C, C, ls1, wp2, tabr, nsr, 2, bp2, tabr, nsr, st1[2], st1[3], True)
        ga2, gug = torch.empty_like(x), torch.empty_like(x)                       # through proj_2 and the gate u * a2
        _gemm(lib, st, w3t, G, C, C, P, N, _E_MUL2, ga2, gug, s=(u, a2))
        tab2, ns2 = _reduce(lib, st, ga2, None, N, C, P, 0)
        part, S = _wgrad(lib, st, ga2, a1, C, C, P, N)
        gwc1, gbc1, _ = fold_rows(part, S, C, C, None, None, tab2, ns2, 2, None, None, 0, None, None, False)
        ga1 = torch.empty_like(x)
        _gemm(lib, st, w2t, ga2, C, C, P, N, _E_NONE, ga1)
        del ga2
        ga0 = torch.empty_like(x)                                                 # through the dilated 7x7 and the 5x5
        _lib.check(lib.rsdet_dwconv2d_backward_data_f32(_p(ga1), _p(wd7), N, C, H, W, 7, 3, _p(ga0), None, None, 0, st),
                   "rsdet_dwconv2d_backward_data_f32")
        gwd7, gbd7 = _dw_bwd_w(lib, st, ga1, a0, wd7, N, C, H, W, 7, 3)
        del ga1
        gt1, gs1 = torch.empty_like(x), torch.empty((C,), **f32)                  # + the gate's share of u, through the GELU
        nb = lib.rsdet_dwconv2d_backward_data_ws_size(N, C, H, W)
        ws = torch.empty((max(nb, 4),), dtype=torch.uint8, device=dev)
        _lib.check(lib.rsdet_dwconv2d_backward_data_act_f32(_p(ga0), _p(wd5), N, C, H, W, 5, 1, _p(gug), _p(t1), _p(gt1),
                                                            _p(gs1), _p(ws), nb, st), "rsdet_dwconv2d_backward_data_act_f32")
        gwd5, gbd5 = _dw_bwd_w(lib, st, ga0, u, wd5, N, C, H, W, 5, 1)
        del ga0, gug
        part, S = _wgrad(lib, st, x, gt1, C, C, P, N)                             # UT (C, C) against the RAW x
        gwp1, gbp1, vec1 = fold_bn(part, S, C, C, w1t, gs1, tabr, nsr, ls1, st1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)                                              # norm1's backward + both identity paths
            _gemm(lib, st, w1t, gt1, C, C, P, N, _E_AFFINE, gx, v=(vec1[2], vec1[3], vec1[4], vec1[5]), s=(G, x))
        c11 = (C, C, 1, 1)
        return (gx, vec1[0], vec1[1], gwp1.view(c11), gbp1, gwd5, gbd5, gwd7, gbd7, gwc1.view(c11), gbc1, gwp2.view(c11), gbp2,
                gls1, vec2[0], vec2[1], gwf1.view(R, C, 1, 1), gbf1, gwd3, gbd3, gwf2.view(C, R, 1, 1), gbf2, gls2, None, None)


def _conv_ok(conv, k, dil=1, groups=1):
    return (conv.kernel_size == (k, k) and conv.stride == (1, 1) and conv.dilation == (dil, dil) and conv.groups == groups
            and conv.bias is not None and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous()
            and conv.padding == (dil * (k - 1) // 2,) * 2 and conv.padding_mode == "zeros")


def _bn_ok(bn):
    return (type(bn) is torch.nn.BatchNorm2d and bn.training and bn.track_running_stats and bn.affine
            and bn.momentum is not None and bn.weight.dtype == torch.float32)


def applies(block, x):
    """Does the one-node form take this models/backbones/van.py Block on this input?  (module docstring)"""
    if not (_ON and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
            and not torch.is_autocast_enabled() and torch.is_grad_enabled()):
        return False
    at, mlp = block.attn, block.mlp
    sg = at.spatial_gating_unit
    if not (_bn_ok(block.norm1) and _bn_ok(block.norm2) and isinstance(at.activation, torch.nn.GELU)
            and at.activation.approximate == "none" and isinstance(mlp.act, torch.nn.GELU) and mlp.act.approximate == "none"):
        return False
    N, C, H, W = x.shape
    R = mlp.fc1.out_channels
    if not (_conv_ok(at.proj_1, 1) and _conv_ok(sg.conv1, 1) and _conv_ok(at.proj_2, 1) and _conv_ok(mlp.fc1, 1)
            and _conv_ok(mlp.fc2, 1) and _conv_ok(sg.conv0, 5, 1, C) and _conv_ok(sg.conv_spatial, 7, 3, C)
            and _conv_ok(mlp.dwconv.dwconv, 3, 1, R)):
        return False
    for p in block.parameters():
        if not p.requires_grad:
            return False
    lib = _lib.load()
    P = H * W
    return bool(lib.rsdet_van_gemm_f32_supported(C, C, P, N) and lib.rsdet_van_gemm_f32_supported(R, C, P, N)
                and lib.rsdet_van_gemm_f32_supported(C, R, P, N) and lib.rsdet_van_wgrad_f32_supported(C, C, P, N)
                and lib.rsdet_van_wgrad_f32_supported(C, R, P, N) and N * R <= 65535 and P % 4 == 0)


def van_block(block, x):
    """``block(x)`` for a Block that applies() accepted."""
    at, mlp, n1, n2 = block.attn, block.mlp, block.norm1, block.norm2
    sg = at.spatial_gating_unit
    dw = mlp.dwconv.dwconv
    bn1 = (n1.running_mean, n1.running_var, n1.num_batches_tracked, float(n1.eps), float(n1.momentum))
    bn2 = (n2.running_mean, n2.running_var, n2.num_batches_tracked, float(n2.eps), float(n2.momentum))
    return _VanBlock.apply(x, n1.weight, n1.bias, at.proj_1.weight, at.proj_1.bias, sg.conv0.weight, sg.conv0.bias,
                           sg.conv_spatial.weight, sg.conv_spatial.bias, sg.conv1.weight, sg.conv1.bias, at.proj_2.weight,
                           at.proj_2.bias, block.layer_scale_1, n2.weight, n2.bias, mlp.fc1.weight, mlp.fc1.bias, dw.weight,
                           dw.bias, mlp.fc2.weight, mlp.fc2.bias, block.layer_scale_2, bn1, bn2)
