"""GPU: the identity Bottleneck as one autograd node (ops/bottleneck.py) and the backward-data GEMM with the next backward
step in its epilogue (csrc/gemm1x1_mfma.hip, rsdet_conv1x1_dgrad_bf16) against plain fp32 torch references of the same
arithmetic.  Floating point: bf16 operands, fp32 accumulation, one rounding per stored tensor -- tolerances are the bf16
resolution (2^-8) of the compared quantity's range, written at each assert."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


def _bf(t):
    return t.bfloat16()


@pytest.mark.parametrize("M,C,O", [(4096, 128, 512), (1000, 256, 1024), (130, 32, 64), (128 * 9, 128, 256),
                                   (16384, 512, 2048), (77, 96, 192)])
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_dgrad_gemm_modes_against_fp32(cuda, M, C, O, mode):
    from rs_detection_amd import _lib
    lib = _lib.load()
    assert lib.rsdet_gemm1x1_mfma_supported(M, C, O)
    g = torch.Generator().manual_seed(M + C + O + mode)
    go = _bf(torch.randn(M, O, generator=g)).to(cuda)
    w = _bf(torch.randn(O, C, generator=g) / O ** 0.5).to(cuda)                  # the convolution's (O, C) weight
    side = _bf(torch.randn(M, C, generator=g)).to(cuda)
    wt = torch.empty((C, O), dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.rsdet_weight_transpose_scale_bf16(_lib.ptr(w), O, C, None, None, 0.0, _lib.ptr(wt), _lib.stream_ptr()), "t")
    assert torch.equal(wt, w.t().contiguous())
    out = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=cuda)
    gb = torch.full((C,), float("nan"), device=cuda)
    nb = lib.rsdet_conv1x1_dgrad_ws_size(M, C, O)
    ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=cuda)
    rc = lib.rsdet_conv1x1_dgrad_bf16(_lib.ptr(go), _lib.ptr(wt), M, C, O, mode, _lib.ptr(side) if mode else None,
                                      _lib.ptr(gb) if mode == 2 else None, _lib.ptr(ws) if mode == 2 else None, nb,
                                      _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv1x1_dgrad_bf16")
    acc = go.float() @ w.float()
    if mode == 0:
        ref = acc
    elif mode == 3:
        ref = acc + side.float()
    else:
        ref = acc * (side.float() > 0).float()             # the gated gradient of the BatchNorm's OUTPUT, unscaled
        ref_gb = ref.sum(0)
        # fp32 sums over M products of bf16-valued operands, in another order than torch's: 1e-4 of the sum's scale
        tol = 1e-4 * float(ref.abs().sum(0).max()) + 1e-5
        assert float((gb - ref_gb).abs().max()) <= tol
        # the per-slice table the multi-BatchNorm finish reads: column 0 sums to grad_beta, column 1 is zero
        S = lib.rsdet_conv1x1_dgrad_slices(M, C, O)
        tab = ws[:C * S * 8].view(torch.float32).view(C, S, 2)
        assert float((tab[:, :, 0].sum(1) - ref_gb).abs().max()) <= tol and float(tab[:, :, 1].abs().max()) == 0.0
    assert not torch.isnan(out.float()).any()
    # one bf16 rounding of the result: 2^-8 of its range (+ the accumulation-order noise of an O-term fp32 sum)
    assert float((out.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-5


def test_weight_transpose_scale_and_rowscale_fold(cuda):
    from rs_detection_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for O, C in ((512, 128), (96, 40), (33, 7)):
        w = _bf(torch.randn(O, C, generator=g)).to(cuda)
        var = torch.empty(O).uniform_(0.5, 2, generator=g).to(cuda)
        gamma = torch.empty(O).uniform_(0.5, 1.5, generator=g).to(cuda)
        out = torch.empty((C, O), dtype=torch.bfloat16, device=cuda)
        _lib.check(lib.rsdet_weight_transpose_scale_bf16(_lib.ptr(w), O, C, _lib.ptr(var), _lib.ptr(gamma), 1e-5,
                                                         _lib.ptr(out), _lib.stream_ptr()), "t")
        ref = (w.float() * (gamma * torch.rsqrt(var + 1e-5))[:, None]).t()
        assert float((out.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
    for S, O, C in ((5, 24, 16), (3, 40, 64), (7, 9, 2048), (2, 6, 1100), (4, 5, 4)):
        part = torch.randn(S, O, C, generator=g).to(cuda)
        var = torch.empty(O).uniform_(0.5, 2, generator=g).to(cuda)
        gamma = torch.empty(O).uniform_(0.5, 1.5, generator=g).to(cuda)
        wgt = _bf(torch.randn(O, C, generator=g)).to(cuda)
        for gm in (gamma, None):
            for bf in (0, 1):
                for dot in (False, True):
                    out = torch.empty((O, C), dtype=torch.bfloat16 if bf else torch.float32, device=cuda)
                    d = torch.full((O,), float("nan"), device=cuda) if dot else None
                    _lib.check(lib.rsdet_sum_slabs_rowscale_f32(_lib.ptr(part), S, O * C, C, _lib.ptr(var), _lib.ptr(gm), 1e-5,
                                                                _lib.ptr(wgt) if dot else None, _lib.ptr(d), _lib.ptr(out), bf,
                                                                _lib.stream_ptr()), "f")
                    acc = part[0].clone()
                    for s in range(1, S):
                        acc += part[s]
                    sc = torch.rsqrt(var + 1e-5) * (1.0 if gm is None else gm)
                    ref = acc * sc[:, None]
                    if bf:
                        assert float((out.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
                    else:
                        assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
                    if dot:         # the UNSCALED fp32 row against the weight: fp32 sum of C products, another order
                        rd = (acc.double() * wgt.double()).sum(1)
                        assert float((d.double() - rd).abs().max()) <= 1e-5 * float((acc.abs() * wgt.float().abs()).sum(1).max())
    # argument checks: a row length that is no multiple of 4; weight without rowdot; mode 2 without its side operand
    assert lib.rsdet_sum_slabs_rowscale_f32(_lib.ptr(part), S, O * C, 6, _lib.ptr(var), None, 1e-5, None, None,
                                            _lib.ptr(out), 1, _lib.stream_ptr()) != 0
    assert lib.rsdet_sum_slabs_rowscale_f32(_lib.ptr(part), S, O * C, C, _lib.ptr(var), None, 1e-5, _lib.ptr(wgt), None,
                                            _lib.ptr(out), 1, _lib.stream_ptr()) != 0
    assert lib.rsdet_conv1x1_dgrad_bf16(_lib.ptr(part), _lib.ptr(part), 128, 48, 64, 2, None, None, None, 0, _lib.ptr(out),
                                        _lib.stream_ptr()) != 0


@pytest.mark.parametrize("B,C,O,H,W", [(2, 128, 128, 40, 36), (1, 64, 256, 33, 70), (4, 256, 8, 16, 16)])
def test_conv3x3_wrw_rowscale_fold_and_row_dots(cuda, B, C, O, H, W):
    """rsdet_conv3x3_wrw_mfma_rowscale_bf16: the scaled weight gradient == scale x the plain entry point's fp32 result, and
    rowdot == sum weight x (that unscaled result) -- to fp32 summation order."""
    from rs_detection_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + C + O)
    assert lib.rsdet_conv3x3_wrw_mfma_supported(B, H, W, C, O)
    cl = torch.channels_last
    go = _bf(torch.randn(B, O, H, W, generator=g)).to(cuda).contiguous(memory_format=cl)
    x = _bf(torch.randn(B, C, H, W, generator=g)).to(cuda).contiguous(memory_format=cl)
    wgt = _bf(torch.randn(O, C, 3, 3, generator=g)).to(cuda).contiguous(memory_format=cl)
    var = torch.empty(O).uniform_(0.5, 2, generator=g).to(cuda)
    gamma = torch.empty(O).uniform_(-1.5, 1.5, generator=g).to(cuda)
    gamma[0] = 0.0
    nb = lib.rsdet_conv3x3_wrw_mfma_ws_size(B, H, W, C, O)
    ws = torch.empty((nb,), dtype=torch.uint8, device=cuda)
    u = torch.empty((O, C, 3, 3), device=cuda).contiguous(memory_format=cl)
    _lib.check(lib.rsdet_conv3x3_wrw_mfma_bf16(_lib.ptr(go), _lib.ptr(x), B, H, W, C, O, _lib.ptr(u), 0, _lib.ptr(ws), nb,
                                               _lib.stream_ptr()), "u")
    sc = gamma * torch.rsqrt(var + 1e-5)
    for bf in (0, 1):
        gw = torch.empty((O, C, 3, 3), dtype=torch.bfloat16 if bf else torch.float32, device=cuda).contiguous(memory_format=cl)
        d = torch.full((O,), float("nan"), device=cuda)
        _lib.check(lib.rsdet_conv3x3_wrw_mfma_rowscale_bf16(_lib.ptr(go), _lib.ptr(x), B, H, W, C, O, _lib.ptr(var),
                                                            _lib.ptr(gamma), 1e-5, _lib.ptr(wgt), _lib.ptr(d), _lib.ptr(gw),
                                                            bf, _lib.ptr(ws), nb, _lib.stream_ptr()), "r")
        ref = u * sc[:, None, None, None]
        if bf:
            assert float((gw.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
        else:                  # (1 / sqrtf in the kernel, rsqrt here: an ulp of the scale)
            assert float((gw - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        rd = (u.double() * wgt.double()).sum((1, 2, 3))
        assert float((d.double() - rd).abs().max()) <= 1e-5 * float((u.abs() * wgt.float().abs()).sum((1, 2, 3)).max())
    assert lib.rsdet_conv3x3_wrw_mfma_rowscale_bf16(_lib.ptr(go), _lib.ptr(x), B, H, W, C, O, _lib.ptr(var), None, 1e-5,
                                                    None, _lib.ptr(d), _lib.ptr(gw), 1, _lib.ptr(ws), nb,
                                                    _lib.stream_ptr()) != 0


def _block(cuda, inplanes, planes, dilation=1, seed=0, small_gamma=False):
    """small_gamma: BatchNorm scales log-uniform in [1e-8, 1e-1] with random signs, some exactly 0, and biases of order 1
    -- the pretrained-ResNet regime (near-dead and zero-initialised residual scales) in which a scale gradient formed from
    xhat = (y - beta) / gamma of the bf16 output is noise; the node must be as accurate there as anywhere."""
    from rs_detection_amd.models.backbones.resnet import Bottleneck
    torch.manual_seed(seed)
    blk = Bottleneck(inplanes, planes, dilation=dilation).to(cuda)
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2), m.running_mean.normal_(0, 0.2), m.running_var.uniform_(0.5, 2)
                if small_gamma:
                    n = m.weight.numel()
                    mag = 10.0 ** torch.empty(n, device=cuda).uniform_(-8, -1)
                    m.weight.copy_(mag * torch.where(torch.rand(n, device=cuda) < 0.3, -1.0, 1.0))
                    m.weight[::7] = 0.0
                    m.bias.normal_(0.5, 0.5)
    return blk.eval()                            # norm_eval: BatchNorm in eval mode, gradients still flow


@pytest.mark.parametrize("small_gamma", [False, True])
@pytest.mark.parametrize("B,inplanes,planes,H,W,dil", [(2, 512, 128, 48, 40, 1), (1, 256, 64, 33, 29, 1),
                                                       (2, 1024, 256, 16, 16, 1), (1, 512, 128, 24, 24, 2)])
def test_one_node_bottleneck_against_the_fp32_composite_under_its_own_gates(cuda, B, inplanes, planes, H, W, dil,
                                                                            small_gamma):
    """Forward: bf16 resolution.  Backward: every gradient against fp32 autograd of the same composite with the ReLU gates
    of the node's own stored activations (elements within a rounding of zero would otherwise flip 100 % of their
    gradient, which measures the rounding of the FORWARD, not the backward under test)."""
    from rs_detection_amd.ops import bottleneck as bt, conv_bn
    blk = _block(cuda, inplanes, planes, dil, small_gamma=small_gamma)
    x = torch.randn(B, inplanes, H, W, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    go = torch.randn(B, inplanes, H, W, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    xi = x.clone().requires_grad_(True)
    assert bt.bottleneck_applies(blk, xi)
    y = blk(xi)
    assert "_Bottleneck" in type(y.grad_fn).__name__
    y.backward(go)
    with torch.no_grad():                        # the node's intermediates, through the same kernels
        from rs_detection_amd.ops.bn_act import bn_act
        y1 = conv_bn.conv_bn_act(blk.conv1, blk.bn1, x)
        y2 = bn_act(F.conv2d(y1, blk.conv2.weight, None, 1, dil, dil), blk.bn2)
    P = {k: v.detach().float().requires_grad_(True) for k, v in blk.named_parameters()}
    xf = x.float().requires_grad_(True)

    def bn(t, n):
        m = getattr(blk, n)
        return F.batch_norm(t, m.running_mean, m.running_var, P[n + ".weight"], P[n + ".bias"], False, 0.0, m.eps)
    a1 = bn(F.conv2d(xf, P["conv1.weight"]), "bn1")
    a1 = a1 * (y1 > 0).float()
    assert float((a1 - y1.float()).abs().max()) <= 2 ** -8 * float(a1.abs().max()) + 1e-6
    z2 = bn(F.conv2d(y1.float() + (a1 - a1.detach()), P["conv2.weight"], None, 1, dil, dil), "bn2")
    a2 = z2 * (y2 > 0).float()
    assert float((a2 - y2.float()).abs().max()) <= 2 * 2 ** -8 * float(a2.abs().max())   # conv2's output + y2: two roundings
    z3 = bn(F.conv2d(y2.float() + (a2 - a2.detach()), P["conv3.weight"]), "bn3") + xf
    a3 = z3 * (y > 0).float()
    assert float((a3 - y.float()).abs().max()) <= 3 * 2 ** -8 * float(a3.abs().max())   # three stored bf16 tensors deep
    a3.backward(go.float())
    assert _rel(xi.grad, xf.grad) <= 2e-2, _rel(xi.grad, xf.grad)
    for k, v in blk.named_parameters():
        tol = 4e-2 if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith("bn3.weight") else 2.5e-2
        assert _rel(v.grad, P[k].grad) <= tol, (k, _rel(v.grad, P[k].grad))


def test_one_node_bottleneck_tracks_the_per_operator_route(cuda):
    from rs_detection_amd.ops import bottleneck as bt
    blk = _block(cuda, 512, 128)
    x = torch.randn(2, 512, 40, 48, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(2, 512, 40, 48, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        bt._ON = on
        try:
            xi = x.clone().requires_grad_(True)
            blk.zero_grad()
            y = blk(xi)
            assert ("_Bottleneck" in type(y.grad_fn).__name__) == on
            y.backward(g)
            outs.append([y.detach().float(), xi.grad.float()] + [p.grad.float() for p in blk.parameters()])
        finally:
            bt._ON = True
    assert _rel(outs[0][0], outs[1][0]) <= 1e-3                  # the same forward arithmetic (bn2 with / without the gate bit mask)
    # gradients: bn2's gate is read from the stored bf16 y2 on both routes; the scale folded into the bf16 weights and the
    # ungated bf16 intermediate the per-operator route stores round differently
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert _rel(a, b) <= 2e-2, _rel(a, b)
    # not taken: a block with a downsample branch, frozen parameters, fp32 input, no grad
    from rs_detection_amd.models.backbones.resnet import Bottleneck
    assert not bt.bottleneck_applies(blk, x.float())
    with torch.no_grad():
        assert not bt.bottleneck_applies(blk, x)
    blk.conv1.weight.requires_grad_(False)
    assert not bt.bottleneck_applies(blk, x)


def test_prepared_weights_follow_every_kind_of_update(cuda):
    """ops/weight_prep.py: one launch prepares every registered operand; torch in-place updates (version counter), our
    fused optimizer (epoch), a reassigned .data (pointer) all make the next request recompute; dead weights drop out."""
    import gc
    from rs_detection_amd.ops import weight_prep as wp
    from rs_detection_amd.ops.conv3x3 import _flipped
    from rs_detection_amd.optims.optimizer import FusedSGD
    g = torch.Generator().manual_seed(3)

    def par(*shape):
        t = torch.randn(*shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
        return torch.nn.Parameter(t)
    w3, w1, wodd = par(96, 64, 3, 3), par(128, 40, 1, 1), par(33, 7, 3, 3)
    var = torch.empty(128).uniform_(0.5, 2, generator=g).to(cuda)
    gamma = torch.nn.Parameter(torch.empty(128).uniform_(0.5, 1.5, generator=g).to(cuda))
    assert wp.applies(w3) and wp.applies(w1) and not wp.applies(w3.detach()) and not wp.applies(w3.float())
    e3, e1, e1s, eo = wp.entry(w3, flip=True), wp.entry(w1), wp.entry(w1, bn=(var, gamma, 1e-5)), wp.entry(wodd, flip=True)
    assert wp.entry(w3, flip=True) is e3

    def check():
        assert torch.equal(e3.tensor(), _flipped(w3.detach()))
        assert torch.equal(eo.tensor(), _flipped(wodd.detach()))
        assert torch.equal(e1.tensor(), w1.detach().reshape(128, 40).t().contiguous())
        ref = (w1.detach().reshape(128, 40).float() * (gamma.detach() * torch.rsqrt(var + 1e-5))[:, None]).t()
        assert float((e1s.tensor().float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
    var3 = torch.empty(96).uniform_(0.5, 2, generator=g).to(cuda)
    gamma3 = torch.nn.Parameter(torch.empty(96).uniform_(-1.5, 1.5, generator=g).to(cuda))
    e3s = wp.entry(w3, bn=(var3, gamma3, 1e-5), flip=True)              # a 3x3 weight with its BatchNorm's scale folded in
    _check0 = check

    def check():
        _check0()
        ref = _flipped((w3.detach().float() * (gamma3.detach() * torch.rsqrt(var3 + 1e-5))[:, None, None, None])
                       .contiguous(memory_format=torch.channels_last))
        assert float((e3s.tensor().float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
    check()
    with torch.no_grad():
        w3.mul_(2.0), w1.add_(1.0)                                     # torch in-place: version counters
    check()
    with torch.no_grad():
        gamma.mul_(0.5)                                                # the BatchNorm's tensors ALONE: theirs too
    check()
    with torch.no_grad():
        var.add_(0.25), var3.mul_(2.0), gamma3.neg_()
    check()
    opt = FusedSGD([w3, w1, wodd, gamma], lr=0.1, momentum=0.9, weight_decay=0.0)
    before = w3.detach().clone()
    for p in (w3, w1, wodd, gamma):
        p.grad = torch.ones_like(p)
    opt.step()                                                         # raw-pointer writes: the epoch
    assert not torch.equal(before, w3.detach())
    check()
    w1.data = (w1.detach() * 0.5).contiguous()                         # a new storage: the pointer
    check()
    reg = e3.reg
    n = len(reg._live())
    del eo, wodd, opt
    gc.collect()
    assert len(reg._live()) == n - 1
    del var3, check, _check0                                           # an entry whose BatchNorm buffer died drops out as well
    gc.collect()
    assert len(reg._live()) == n - 2
    with torch.no_grad():
        w3.mul_(0.5)
    check_alive = torch.equal(e3.tensor(), _flipped(w3.detach()))      # the table was rebuilt without the dead entry
    assert check_alive
