"""GPU: the fused elementwise tails of the VAN block (csrc/van_ops.hip, ops/van_fused.py) against the torch expressions
they replace (the reference's Mlp / LKA / Attention / Block, /root/reference/python/jdet/models/backbones/van.py:46-122)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 32, 48), (1, 320, 8, 8), (3, 10, 5, 7), (2, 512, 3, 2), (1, 4, 129, 130)]


def _close(a, b, tol=2e-6):
    scale = float(b.abs().max()) + 1e-12
    assert a.shape == b.shape and float((a - b).abs().max()) <= tol * scale + 1e-7, float((a - b).abs().max())


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_bias_gelu(cuda, shape, with_bias):
    from rs_detection_amd.ops import van_fused
    torch.manual_seed(shape[1])
    x = (torch.randn(shape, device=cuda) * 2).requires_grad_(True)
    b = torch.randn(shape[1], device=cuda).requires_grad_(True) if with_bias else None
    g = torch.randn(shape, device=cuda)
    assert van_fused.applies(x)
    y = van_fused.bias_gelu(x, b)
    y.backward(g)
    x2 = x.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True) if with_bias else None
    y2 = F.gelu(x2 + b2[None, :, None, None] if with_bias else x2)
    y2.backward(g)
    _close(y.detach(), y2.detach()), _close(x.grad, x2.grad)
    if with_bias:
        _close(b.grad, b2.grad, 2e-5)            # a sum of N*H*W terms in another order


@pytest.mark.parametrize("shape", SHAPES)
def test_gate(cuda, shape):
    from rs_detection_amd.ops import van_fused
    torch.manual_seed(shape[2])
    u, a = torch.randn(shape, device=cuda).requires_grad_(True), torch.randn(shape, device=cuda).requires_grad_(True)
    b = torch.randn(shape[1], device=cuda).requires_grad_(True)
    g = torch.randn(shape, device=cuda)
    y = van_fused.gate(u, a, b)
    y.backward(g)
    u2, a2, b2 = (t.detach().clone().requires_grad_(True) for t in (u, a, b))
    y2 = u2 * (a2 + b2[None, :, None, None])
    y2.backward(g)
    _close(y.detach(), y2.detach()), _close(u.grad, u2.grad), _close(a.grad, a2.grad), _close(b.grad, b2.grad, 2e-5)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("with_shortcut,with_bias", [(True, True), (False, True), (False, False)])
def test_residual(cuda, shape, with_shortcut, with_bias):
    """x + ls * (p + b + shortcut): values, and the five gradients -- x's is the incoming gradient itself, the shortcut's
    and p's are the same ls * g, the bias' and the layer scale's are per-channel sums."""
    from rs_detection_amd.ops import van_fused
    torch.manual_seed(shape[3])
    C = shape[1]
    x, p = torch.randn(shape, device=cuda).requires_grad_(True), torch.randn(shape, device=cuda).requires_grad_(True)
    sc = torch.randn(shape, device=cuda).requires_grad_(True) if with_shortcut else None
    b = torch.randn(C, device=cuda).requires_grad_(True) if with_bias else None
    ls = (0.01 + torch.rand(C, device=cuda)).requires_grad_(True)
    g = torch.randn(shape, device=cuda)
    y = van_fused.residual(x, p, b, sc, ls)
    y.backward(g)
    x2, p2, ls2 = (t.detach().clone().requires_grad_(True) for t in (x, p, ls))
    sc2 = sc.detach().clone().requires_grad_(True) if with_shortcut else None
    b2 = b.detach().clone().requires_grad_(True) if with_bias else None
    f = p2
    if with_bias:
        f = f + b2[None, :, None, None]
    if with_shortcut:
        f = f + sc2
    y2 = x2 + ls2[None, :, None, None] * f
    y2.backward(g)
    _close(y.detach(), y2.detach()), _close(x.grad, x2.grad, 0.0), _close(p.grad, p2.grad), _close(ls.grad, ls2.grad, 2e-5)
    if with_shortcut:
        _close(sc.grad, sc2.grad)
    if with_bias:
        _close(b.grad, b2.grad, 2e-5)


def test_van_block_fused_equals_unfused(cuda, monkeypatch):
    """A whole Block (and a two-block stage, so that gradients shared between the fused passes accumulate): the fused
    tails against RSDET_VAN_FUSED=0 -- outputs, input gradient, every parameter gradient."""
    from rs_detection_amd.models.backbones.van import Block
    from rs_detection_amd.ops import van_fused
    torch.manual_seed(3)
    blocks = torch.nn.Sequential(Block(64, mlp_ratio=8), Block(64, mlp_ratio=8)).to(cuda).train()
    with torch.no_grad():
        for blk in blocks:
            blk.layer_scale_1.uniform_(0.2, 1.0), blk.layer_scale_2.uniform_(0.2, 1.0)
            for m in blk.modules():
                if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
                    m.bias.normal_(0, 0.2)
    x = torch.randn((2, 64, 24, 40), device=cuda)
    g = torch.randn_like(x)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(van_fused, "_ON", on)
        xi = x.clone().requires_grad_(True)
        blocks.zero_grad()
        for blk in blocks:                      # same batch statistics both times: BatchNorm in training mode is stateful
            blk.norm1.momentum = blk.norm2.momentum = 0.0
        y = blocks(xi)
        if on:
            assert "_Residual" in type(y.grad_fn).__name__
        y.backward(g)
        outs.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in blocks.parameters()])
    names = ["y", "gx"] + [n for n, _ in blocks.named_parameters()]
    for n, a, b in zip(names, *outs):
        assert float((a - b).abs().max()) <= 2e-4 * (float(b.abs().max()) + 1e-6), n


@pytest.mark.parametrize("C,O,H,W", [(64, 512, 128, 128), (128, 64, 128, 256), (64, 64, 256, 256), (320, 1280, 64, 64),
                                     (512, 512, 32, 32), (24, 40, 7, 9)])
def test_conv1x1_nchw_split_k_weight_gradient(cuda, C, O, H, W):
    """ops/conv1x1.conv1x1_nchw: forward and input gradient are batched GEMMs on views of the NCHW maps (round 5), the
    weight gradient a split-K batched GEMM on large maps and MIOpen's kernel on small ones -- against
    nn.functional.conv2d's own output and gradients (fp32 sums in another order: 1e-5 relative)."""
    from rs_detection_amd.ops import conv1x1 as c1
    torch.manual_seed(C + O)
    x = torch.randn((2, C, H, W), device=cuda)
    w = torch.randn((O, C, 1, 1), device=cuda) * 0.05
    g = torch.randn((2, O, H, W), device=cuda)
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    c1._NCHW_MM = True                           # the GEMM-on-views route (measured, not the default): every map size
    try:
        y = c1.conv1x1_nchw(xa, wa)
        assert "Conv1x1NCHW" in type(y.grad_fn).__name__
        y.backward(g)
    finally:
        c1._NCHW_MM = False
    xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y2 = F.conv2d(xb, wb)
    y2.backward(g)
    _close(y.detach(), y2.detach(), 1e-5)
    _close(xa.grad, xb.grad, 1e-5)
    _close(wa.grad, wb.grad, 2e-5)
    # the default route: MIOpen's own forward / backward-data, our split-K weight gradient on large maps only
    if H * W >= c1._NCHW_WRW_MIN_PIXELS and (H * W) % (c1._NCHW_WRW_SPLITS * 64) == 0:
        xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y3 = c1.conv1x1_nchw(xc, wc)
        assert "Conv1x1NCHW" in type(y3.grad_fn).__name__ and torch.equal(y3.detach(), y2.detach())
        y3.backward(g)
        _close(xc.grad, xb.grad, 1e-6)
        _close(wc.grad, wb.grad, 2e-5)
    else:
        assert "Conv1x1NCHW" not in type(c1.conv1x1_nchw(xa, wa).grad_fn).__name__
