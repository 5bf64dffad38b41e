"""GPU: a VAN Block as one autograd node (ops/van_block.py: fp32 MFMA GEMMs with fused tails, csrc/van_gemm.hip) and its
C-ABI pieces against plain torch references -- the block in float64 through torch's own operators (every fused op of the
package falls back to the torch expression for float64), i.e. autograd of the composite the node replaces.  The node is
exact fp32: output, all 23 gradients and the updated running statistics agree to <= 1e-4 relative (measured ~1e-6)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _block(cuda, dim, ratio, seed):
    from rs_detection_amd.models.backbones.van import Block
    torch.manual_seed(seed)
    blk = Block(dim, mlp_ratio=ratio).to(cuda)
    with torch.no_grad():            # away from the initialisation's symmetric points: every term of every gradient is live
        for n, p in blk.named_parameters():
            if "layer_scale" in n:
                p.uniform_(0.05, 0.5)
            elif "norm" in n and n.endswith("weight"):
                p.uniform_(0.5, 1.5)
            elif n.endswith("bias"):
                p.normal_(0, 0.2)
            elif p.dim() == 4 and p.shape[1] == 1:
                p.normal_(0, 0.3 / p.shape[-1])
            else:
                p.normal_(0, 1.0 / p.shape[1] ** 0.5)
        for bn in (blk.norm1, blk.norm2):
            bn.running_mean.normal_(0, 0.1), bn.running_var.uniform_(0.5, 2)
    return blk.train()


@pytest.mark.parametrize("N,dim,ratio,H,W", [(2, 64, 8, 32, 32), (1, 128, 8, 16, 32), (2, 320, 4, 16, 16), (1, 512, 4, 8, 16),
                                             (2, 64, 8, 128, 128)])
def test_one_node_block_against_float64_autograd_of_the_composite(cuda, N, dim, ratio, H, W):
    from rs_detection_amd.ops import van_block as vb
    blk = _block(cuda, dim, ratio, N + dim)
    ref = copy.deepcopy(blk).double()
    x = (torch.randn(N, dim, H, W, device=cuda) * 1.5 + 0.3)
    go = torch.randn(N, dim, H, W, device=cuda)
    xi = x.clone().requires_grad_(True)
    assert vb.applies(blk, xi)
    y = blk(xi)
    assert "_VanBlock" in type(y.grad_fn).__name__
    y.backward(go)
    xd = x.double().requires_grad_(True)
    yd = ref(xd)
    assert "_VanBlock" not in type(yd.grad_fn).__name__
    yd.backward(go.double())
    assert _rel(y, yd) <= 1e-5, _rel(y, yd)
    assert _rel(xi.grad, xd.grad) <= 1e-4, _rel(xi.grad, xd.grad)
    for (n, p), q in zip(blk.named_parameters(), ref.parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, n
        assert _rel(p.grad, q.grad) <= 1e-4, (n, _rel(p.grad, q.grad))
    for bn, bd in ((blk.norm1, ref.norm1), (blk.norm2, ref.norm2)):      # nn.BatchNorm2d's running-statistics update
        assert _rel(bn.running_mean, bd.running_mean) <= 1e-5 and _rel(bn.running_var, bd.running_var) <= 1e-5
        assert int(bn.num_batches_tracked) == int(bd.num_batches_tracked) == 1


def test_block_node_tracks_the_per_operator_route_and_is_not_taken_where_it_does_not_apply(cuda):
    from rs_detection_amd.ops import van_block as vb
    blk = _block(cuda, 64, 8, 5)
    x = torch.randn(2, 64, 64, 64, device=cuda)
    g = torch.randn(2, 64, 64, 64, device=cuda)
    outs = []
    for on in (True, False):
        vb._ON = on
        try:
            b2 = copy.deepcopy(blk)
            xi = x.clone().requires_grad_(True)
            y = b2(xi)
            assert ("_VanBlock" in type(y.grad_fn).__name__) == on
            y.backward(g)
            outs.append([y.detach(), xi.grad] + [p.grad for p in b2.parameters()] + [b2.norm1.running_var, b2.norm2.running_mean])
        finally:
            vb._ON = True
    for a, b in zip(*outs):
        assert _rel(a, b) <= 1e-4, _rel(a, b)
    # not taken: eval-mode BatchNorm, no grad, autocast, a frozen parameter, a map size the tiles do not divide
    assert not vb.applies(copy.deepcopy(blk).eval(), x)
    with torch.no_grad():
        assert not vb.applies(blk, x)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not vb.applies(blk, x)
    assert not vb.applies(blk, torch.randn(2, 64, 30, 30, device=cuda))
    b3 = copy.deepcopy(blk)
    b3.attn.proj_1.bias.requires_grad_(False)
    assert not vb.applies(b3, x)
    y = blk(torch.randn(2, 64, 30, 30, device=cuda, requires_grad=True))     # the per-operator route still serves it
    assert "_VanBlock" not in type(y.grad_fn).__name__


def test_side_stream_backward_is_bit_identical_to_the_one_stream_order(cuda):
    """The backward's off-chain work (weight gradients, folds, depthwise weight gradients) on a side stream: the same
    kernels on the same operands -- every gradient bit-equal to the one-stream order, over repeated steps (a missing
    cross-stream edge would show as a race sooner or later)."""
    from rs_detection_amd import _lib
    lib = _lib.load()
    blk = _block(cuda, 320, 4, 11)
    x = torch.randn(2, 320, 32, 32, device=cuda)
    go = torch.randn(2, 320, 32, 32, device=cuda)
    res = {}
    prev = lib.rsdet_van_block_side_stream(-1)
    try:
        for on in (0, 1):
            lib.rsdet_van_block_side_stream(on)
            outs = []
            for it in range(6):
                b2 = copy.deepcopy(blk)
                xi = x.clone().requires_grad_(True)
                y = b2(xi)
                assert "_VanBlock" in type(y.grad_fn).__name__
                y.backward(go)
                outs.append([xi.grad.clone()] + [p.grad.clone() for p in b2.parameters()])
            torch.cuda.synchronize()
            for o in outs[1:]:
                assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
            res[on] = outs[0]
    finally:
        lib.rsdet_van_block_side_stream(prev)
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))


@pytest.mark.parametrize("M,Nn,P,n", [(64, 64, 4096, 2), (320, 1280, 256, 2), (128, 1024, 512, 1), (512, 512, 128, 3)])
def test_wgrad_and_folds_through_the_c_abi(cuda, M, Nn, P, n):
    """rsdet_van_wgrad_f32 + rsdet_van_fold_rows_f32 against float64 einsum; the row dots, bias and scale gradients."""
    import ctypes
    from rs_detection_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + Nn)
    g = torch.randn(n, M, P, generator=gen).to(cuda)
    x = torch.randn(n, Nn, P, generator=gen).to(cuda)
    w = torch.randn(M, Nn, generator=gen).to(cuda)
    rs, bias = torch.randn(M, generator=gen).to(cuda), torch.randn(M, generator=gen).to(cuda)
    assert lib.rsdet_van_wgrad_f32_supported(M, Nn, P, n)
    S = lib.rsdet_van_wgrad_f32_splits(M, Nn, P, n)
    part = torch.empty((S, M, Nn), device=cuda)
    _lib.check(lib.rsdet_van_wgrad_f32(_lib.ptr(g), _lib.ptr(x), M, Nn, P, n, _lib.ptr(part), _lib.stream_ptr()), "w")
    U = torch.einsum("nmp,nkp->mk", g.double(), x.double())
    assert _rel(part.double().sum(0), U) <= 1e-6
    ns = n * lib.rsdet_van_chan_slices(P)
    tab = torch.empty((M, ns, 2), device=cuda)
    _lib.check(lib.rsdet_van_chan_reduce_f32(_lib.ptr(g), None, n, M, P, 0, _lib.ptr(tab), _lib.stream_ptr()), "r")
    gs = g.double().sum((0, 2))
    assert _rel(tab[:, :, 0].double().sum(1), gs) <= 1e-6 and float(tab[:, :, 1].abs().max()) == 0.0
    gw, gb, grs = torch.empty((M, Nn), device=cuda), torch.empty(M, device=cuda), torch.empty(M, device=cuda)
    f = _lib.VanRowsFold(part.data_ptr(), rs.data_ptr(), w.data_ptr(), tab.data_ptr(), bias.data_ptr(), None, None, None,
                         gw.data_ptr(), gb.data_ptr(), grs.data_ptr(), S, M, Nn, ns, 2, 0)
    _lib.check(lib.rsdet_van_fold_rows_f32(ctypes.byref(f), _lib.stream_ptr()), "f")
    assert _rel(gw, U * rs.double()[:, None]) <= 1e-6 and _rel(gb, gs * rs.double()) <= 1e-6
    assert _rel(grs, (U * w.double()).sum(1) + bias.double() * gs) <= 1e-5
    # unsupported geometry is refused, not mangled
    assert not lib.rsdet_van_wgrad_f32_supported(96, 64, 256, 1) and not lib.rsdet_van_gemm_f32_supported(320, 48, 256, 1)
    assert lib.rsdet_van_gemm_f32(_lib.ptr(w), _lib.ptr(x), 96, 64, 256, 1, 0, None, None, None, None, None, None,
                                  _lib.ptr(part), None, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("M,K,P,n", [(320, 320, 256, 2), (64, 512, 1024, 1), (1024, 128, 512, 2), (512, 2048, 128, 1)])
def test_gemm_epilogues_through_the_c_abi(cuda, M, K, P, n):
    """rsdet_van_gemm_f32, every epilogue, against float64 matmul of the same fp32 operands."""
    import torch.nn.functional as F
    from rs_detection_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + K + P)
    w = (torch.randn(M, K, generator=gen) / K ** 0.5).to(cuda)
    x = torch.randn(n, K, P, generator=gen).to(cuda)
    v = [torch.randn(M, generator=gen).to(cuda) for _ in range(4)]
    s0, s1 = torch.randn(n, M, P, generator=gen).to(cuda), torch.randn(n, M, P, generator=gen).to(cuda)

    def run(epi, vv=(None,) * 4, ss=(None, None), two=False):
        o0 = torch.full((n, M, P), float("nan"), device=cuda)
        o1 = torch.full((n, M, P), float("nan"), device=cuda) if two else None
        rc = lib.rsdet_van_gemm_f32(_lib.ptr(w), _lib.ptr(x), M, K, P, n, epi, *[_lib.ptr(t) for t in vv],
                                    *[_lib.ptr(t) for t in ss], _lib.ptr(o0), _lib.ptr(o1), _lib.stream_ptr())
        _lib.check(rc, "rsdet_van_gemm_f32")
        return o0, o1
    ref = torch.matmul(w.double(), x.double())
    col = lambda t: t.double()[None, :, None]
    tol = 2e-6
    assert _rel(run(0)[0], ref) <= tol
    assert _rel(run(1, (v[0], None, None, None))[0], ref + col(v[0])) <= tol
    o0, o1 = run(2, (v[0], None, None, None), two=True)
    assert _rel(o0, ref + col(v[0])) <= tol and _rel(o1, F.gelu(ref + col(v[0]))) <= tol
    o0, o1 = run(3, (v[0], None, None, None), (s0, None), two=True)
    assert _rel(o0, ref + col(v[0])) <= tol and _rel(o1, (ref + col(v[0])) * s0.double()) <= tol
    assert _rel(run(4, v, (s0, s1))[0], s0.double() * col(v[0]) + ref * col(v[1]) + col(v[2]) + s1.double() * col(v[3])) <= tol
    assert _rel(run(4, (None, v[1], v[2], None), (s0, None))[0], s0.double() + ref * col(v[1]) + col(v[2])) <= tol
    o0, o1 = run(5, ss=(s0, s1), two=True)
    assert _rel(o0, ref * s0.double()) <= tol and _rel(o1, ref * s1.double()) <= tol
    assert _rel(run(6, ss=(s0, None))[0], ref * s0.double()) <= tol


def test_merged_folds_and_depthwise_finishes_equal_the_single_calls(cuda):
    """rsdet_van_fold_rows_multi_f32 (three folds in one launch) == three rsdet_van_fold_rows_f32 calls, and
    rsdet_dwconv2d_backward_weight_partial_f32 + rsdet_dwconv2d_wgrad_finish_multi_f32 == rsdet_dwconv2d_backward_weight_f32,
    bit for bit (same summation order)."""
    import ctypes
    from rs_detection_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(5)
    jobs, singles = (_lib.VanRowsFold * 3)(), []
    keep = []
    for j, (S, M, Nn, with_dot) in enumerate(((6, 320, 1280, True), (25, 320, 320, True), (25, 64, 64, False))):
        part = torch.randn(S, M, Nn, generator=gen).to(cuda)
        w, rs, bias = (torch.randn(*sh, generator=gen).to(cuda) for sh in ((M, Nn), (M,), (M,)))
        tab = torch.randn(M, 4, 2, generator=gen).to(cuda)
        outs = [[torch.empty((M, Nn), device=cuda), torch.empty(M, device=cuda), torch.empty(M, device=cuda)] for _ in range(2)]
        for which in range(2):
            gw, gb, grs = outs[which]
            f = _lib.VanRowsFold(part.data_ptr(), rs.data_ptr(), w.data_ptr() if with_dot else None, tab.data_ptr(),
                                 bias.data_ptr(), None, None, None, gw.data_ptr(), gb.data_ptr(),
                                 grs.data_ptr() if with_dot else None, S, M, Nn, 4, 2, 0)
            if which == 0:
                _lib.check(lib.rsdet_van_fold_rows_f32(ctypes.byref(f), _lib.stream_ptr()), "single")
            else:
                jobs[j] = f
        keep.append((part, w, rs, bias, tab))
        singles.append((outs, with_dot))
    _lib.check(lib.rsdet_van_fold_rows_multi_f32(jobs, 3, _lib.stream_ptr()), "multi")
    for outs, with_dot in singles:
        for a, b in list(zip(outs[0], outs[1]))[:3 if with_dot else 2]:
            assert torch.equal(a, b)
    assert lib.rsdet_van_fold_rows_multi_f32(jobs, 4, _lib.stream_ptr()) != 0
    # depthwise weight gradients
    N, H, W = 2, 64, 64
    cfgs = ((96, 3), (32, 7), (32, 5))
    ws, one, multi = [], [], []
    P = _lib.ptr
    for C, K in cfgs:
        gy, x = torch.randn(N, C, H, W, generator=gen).to(cuda), torch.randn(N, C, H, W, generator=gen).to(cuda)
        nb = lib.rsdet_dwconv2d_backward_weight_ws_size(N, C, H, W, K)
        w1, w2 = torch.empty(nb, dtype=torch.uint8, device=cuda), torch.empty(nb, dtype=torch.uint8, device=cuda)
        gw1, gb1 = torch.empty(C, 1, K, K, device=cuda), torch.empty(C, device=cuda)
        gw2, gb2 = torch.empty(C, 1, K, K, device=cuda), torch.empty(C, device=cuda)
        D = 3 if K == 7 else 1
        _lib.check(lib.rsdet_dwconv2d_backward_weight_f32(P(gy), P(x), None, N, C, H, W, K, D, P(gw1), P(gb1), P(w1), nb,
                                                          _lib.stream_ptr()), "one call")
        _lib.check(lib.rsdet_dwconv2d_backward_weight_partial_f32(P(gy), P(x), None, N, C, H, W, K, D, P(w2), nb,
                                                                  _lib.stream_ptr()), "partial")
        ws.append(w2), one.append((gw1, gb1)), multi.append((gw2, gb2))
        keep.append((gy, x, w1))
    arr = lambda ty, vals: (ty * 3)(*vals)
    rc = lib.rsdet_dwconv2d_wgrad_finish_multi_f32(
        3, arr(ctypes.c_void_p, [w.data_ptr() for w in ws]), arr(ctypes.c_int, [N] * 3), arr(ctypes.c_int, [c for c, _ in cfgs]),
        arr(ctypes.c_int, [H] * 3), arr(ctypes.c_int, [W] * 3), arr(ctypes.c_int, [k for _, k in cfgs]),
        arr(ctypes.c_void_p, [m[0].data_ptr() for m in multi]), arr(ctypes.c_void_p, [m[1].data_ptr() for m in multi]),
        _lib.stream_ptr())
    _lib.check(rc, "finish multi")
    for (a, b), (c, d) in zip(one, multi):
        assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 64, 64), (1, 128, 24, 20), (2, 320, 16, 16), (1, 512, 9, 7), (1, 20, 5, 3)])
def test_channel_layernorm_on_the_nchw_map_equals_the_tensor_form(cuda, N, C, H, W):
    """rsdet_chan_layernorm_* (the norm at the end of a VAN stage on the NCHW map) == flatten / transpose / nn.LayerNorm /
    permute back: output, input gradient and both parameter gradients (float64 composite as the reference)."""
    from rs_detection_amd.ops import chan_layernorm
    torch.manual_seed(C)
    norm = torch.nn.LayerNorm(C).to(cuda)
    with torch.no_grad():
        norm.weight.normal_(1, 0.3), norm.bias.normal_(0, 0.3)
    x = (torch.randn(N, C, H, W, device=cuda) * 2 + 0.5).requires_grad_(True)
    go = torch.randn(N, C, H, W, device=cuda)
    assert chan_layernorm.applies(x, norm)
    y = chan_layernorm.chan_layer_norm(x, norm)
    gx, gw, gb = torch.autograd.grad(y, (x, norm.weight, norm.bias), go)
    xd = x.detach().double().requires_grad_(True)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xd.flatten(2).transpose(1, 2), (C,), wd, bd, norm.eps)
    yr = yr.reshape(N, H, W, C).permute(0, 3, 1, 2)
    rx, rw, rb = torch.autograd.grad(yr, (xd, wd, bd), go.double())
    for a, b, tol in ((y, yr, 2e-6), (gx, rx, 2e-5), (gw, rw, 2e-5), (gb, rb, 2e-5)):
        assert _rel(a, b) <= tol, (_rel(a, b), tol)
    assert not chan_layernorm.applies(x.detach().half(), norm)
